/*
 * hmmufotu_amd — C ABI of the MI355X-native per-read assignment engine.
 *
 * The reference (HmmUFOtu v1.5.1) has no plugin/FFI layer; its seam for this path is the C++
 * free-function API of src/HmmUFOtu_main.h:70-113 called once per read from the OpenMP task in
 * src/hmmufotu.cpp:603-751.  This header is the batched, POD-only replacement of that seam:
 * every entry point names the reference function(s) it stands in for.  Host code that keeps
 * the per-read reference signatures lives in hmmufotu_amd/csrc/hu_reference_api.hpp.
 *
 * Conventions: all pointers are HOST pointers unless a field says "device"; the caller owns
 * every in/out buffer; functions return HU_OK (0) or a negative hu_status and never abort;
 * hu_last_error() returns a thread-local message.  One hu_batch is driven by one host thread
 * (it owns a HIP stream); several batches may share one hu_db concurrently (the DB is
 * immutable after creation, like the const BandedHMMP7 / PTUnrooted objects of the reference).
 */
#ifndef HMMUFOTU_AMD_H_
#define HMMUFOTU_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
	HU_OK = 0,
	HU_ERR_ARG = -1,      /* invalid argument (reference: assert / invalid_argument)            */
	HU_ERR_DEVICE = -2,   /* no gfx950 device / HIP failure: the engine has NO CPU fallback      */
	HU_ERR_IO = -3,       /* unreadable or malformed database file                                */
	HU_ERR_NOMEM = -4,
	HU_ERR_STATE = -5     /* stage called out of order                                            */
} hu_status;

/* align_mode of src/BandedHMMP7.h:168-173 */
enum { HU_MODE_GLOBAL = 0, HU_MODE_LOCAL = 1, HU_MODE_NGCL = 2, HU_MODE_CGNL = 3 };
/* DNA substitution models of src/DNASubModelFactory.cpp:38-53 */
enum { HU_GTR = 0, HU_TN93 = 1, HU_HKY85 = 2, HU_F81 = 3, HU_K80 = 4, HU_JC69 = 5 };
/* PTUnrooted::PRIOR_TYPE */
enum { HU_PRIOR_UNIFORM = 0, HU_PRIOR_HEIGHT = 1 };

/* per-read status flags (replace the reference's assert(aln.isValid()) aborts, src/hmmufotu.cpp:622) */
enum {
	HU_READ_OK = 1,
	HU_READ_INVALID = 0,      /* bad base code or no finite Viterbi path                            */
	HU_READ_CHIMERA = 2,      /* PE orientation check failed (src/hmmufotu.cpp:629-637)             */
	HU_READ_NEEDS_FULL = 4,   /* internal: banded DP found no path, full DP scheduled              */
	HU_READ_OUT_OF_WINDOW = 16 /* aligned, but its CS region leaves the column window the database keeps messages for
	                            * (hu_tree_desc.win_*): not placed, no TSV line; the rest of the batch goes on   */
};

/* capacity of the per-read seed lists: max_nseed <= HU_MAX_SEEDS, and the per-seed result arrays of
 * hu_batch_get_seeds / hu_batch_get_estimates / hu_seed_batch_given have this ROW STRIDE whatever max_nseed is */
#define HU_MAX_SEEDS 64

typedef struct hu_db hu_db;        /* profile + pre-evaluated tree packed once into HBM             */
typedef struct hu_batch hu_batch;  /* device workspace + stream for one batch of reads in flight    */

/* BandedHMMP7 profile fields as operator>> leaves them (src/BandedHMMP7.cpp:100-246,
 * src/BandedHMMP7.h:499-548): costs are -ln p, '*' == +inf.  hu_db_create runs the post-load
 * chain itself (extend_index, adjustProfileLocalMode, wingRetract; :104-109, :1083-1120). */
typedef struct {
	int32_t K;                /* profile length (LENG)                                             */
	int32_t L;                /* consensus length (MAXL)                                           */
	const double* EM;         /* [K+1][4] match emission costs, row 0 = COMPO                      */
	const double* EI;         /* [K+1][4] insert emission costs                                    */
	const double* T;          /* [K+1][7] transition costs m->m m->i m->d i->m i->i d->m d->d      */
	const int32_t* p2cs;      /* [K+1] MAP: 1-based CS column of profile column k; [0] ignored     */
} hu_profile_desc;

/* DNASubModel + DiscreteGammaModel parameters (src/GTR.cpp:43-81, src/TN93.cpp:40-77, ...;
 * src/DiscreteGammaModel.cpp:57-72).  par: GTR R[16] row-major | TN93 kr,ky,beta |
 * HKY85 kappa,beta | F81 beta | K80 kappa | JC69 -.  pi is ignored for K80/JC69. */
typedef struct {
	int32_t type;
	double pi[4];
	double par[16];
	int32_t dg_k;             /* 0 = no discrete Gamma                                             */
	double dg_rate[16];       /* r[k] exactly as stored in the .ptu (NOT multiplied by K)          */
} hu_model_desc;

/* PTUnrooted as loaded from a .ptu (src/PhyloTreeUnrooted.cpp:496-535): node i is id2node[i].
 * up[i] is the cached message of branch i -> parent(i), down[i] that of parent(i) -> i
 * (Matrix4Xd 4 x csLen, column-major == [site][4]); for the root, up[root] holds the root
 * message.  Messages may be given for a column window only (win_len < cs_len). */
typedef struct {
	int32_t n_nodes;
	int32_t cs_len;
	const int32_t* parent;    /* [n] -1 for the root                                               */
	const double* blen;       /* [n] length of the branch to the parent                            */
	const int8_t* seq;        /* [n][cs_len] DigitalSeq codes: 0..3, gap -2                        */
	const double* up;         /* [n][win_len][4]                                                   */
	const double* down;       /* [n][win_len][4]                                                   */
	const double* height;     /* [n] node2height                                                   */
	const int32_t* anno_id;   /* [n] class of the node's taxon annotation string (may be NULL)     */
	const double* anno_dist;  /* [n] annoDist (may be NULL)                                        */
	int64_t win_start;        /* first CS column (0-based) covered by up/down                      */
	int64_t win_len;          /* 0 = all cs_len columns                                            */
	int32_t msgs_on_device;   /* up/down are DEVICE pointers the DB adopts: not copied, not freed, and
	                           * REWRITTEN IN PLACE into the engine's packed linear form             */
} hu_tree_desc;

/* CLI defaults of src/hmmufotu.cpp:37-57 */
typedef struct {
	int32_t align_mode;       /* HU_MODE_GLOBAL for assembled/PE reads, HU_MODE_NGCL for --single  */
	int32_t max_nseed;        /* -N  [50]                                                          */
	double max_diff;          /* -d  [inf]                                                         */
	double max_height;        /* -H  [inf]                                                         */
	double max_error;         /* -e  [20]                                                          */
	int32_t weighted;         /* -m  0 'unweighted' [default], 1 'weighted'                        */
	int32_t only_ml;          /* --ML                                                              */
	int32_t prior;            /* --prior                                                           */
	int32_t ignore_orient;    /* -i                                                                */
	int32_t fix_root_loglik;  /* 0 [default]: placeSeq's loglik as the reference computes it — the constant
	                           * (end - start + 1) log(sum_i pi_i e) of SURVEY.md F4 (src/PhyloTreeUnrooted.cpp:918-922), so every
	                           * candidate ties and the pick follows the estimated order.  1 (--fix-root-loglik): the value evidently
	                           * intended, sum_j log pi . exp(loglik(r, j)) at the optimised branch lengths; candidates are then
	                           * ranked by it (q-values, --ML sort, chimera log-odds all become informative).  A documented
	                           * deviation from the reference, off by default                                          */
	int32_t seed_order;       /* which of the nodes that TIE at the cut-off distance getSeed keeps, and in which order the seeds reach estimateSeq:
	                           * HU_SEED_ORDER_LIBSTDCXX (1, the default of hu_default_opts): the reference's own — the first max_nseed elements of
	                           * std::sort(locs) on dist ALONE (src/HmmUFOtu_main.cpp:139, src/hmmufotu.cpp:646-647), i.e. the tie permutation of
	                           * libstdc++'s introsort over all ~n_nodes PTLocs in node order.  Reproduced ON THE DEVICE (full (d, N) pair scan +
	                           * k_seed_refsort: data-parallel Hoare partitions restricted to the ranges that reach the first max_nseed places).
	                           * The host restatement (hu_sort_prefix_libstdcxx) finishes only the reads the kernel lists: a row that reaches
	                           * introsort's heap-sort branch, level tables that overflow, and every read of a tree whose sort tables exceed
	                           * 150 KB of LDS (its index of every 64th table entry: above ~9 M nodes; reported once on stderr and in hu_batch_refsort_stats).  A read with a NaN
	                           * distance (a node sharing no column with it: std::sort is undefined) takes (dist, node id) with NaN last.
	                           * HU_SEED_ORDER_STABLE (0): ascending (dist, node id) — independent of any library's tie permutation, selected by
	                           * the distance-only scan + top-k; ~25 % faster, differs from a g++-built reference wherever nodes tie at the
	                           * cut-off distance.  NOTE: a zero-filled hu_opts selects STABLE; call hu_default_opts. */
} hu_opts;
enum { HU_SEED_ORDER_STABLE = 0, HU_SEED_ORDER_LIBSTDCXX = 1 };

/* BandedHMMP7::HmmAlignment minus the string (src/BandedHMMP7.h:74-130) */
typedef struct {
	int32_t seq_start, seq_end, hmm_start, hmm_end, cs_start, cs_end; /* all 1-based */
	int32_t status;           /* HU_READ_*                                                          */
	int32_t used_full;        /* 1 if the full-DP fallback ran (src/HmmUFOtu_main.cpp:89-93)        */
	double cost;
} hu_align_rec;

/* PTUnrooted::PTPlacement (src/PhyloTreeUnrooted.h:410-510) with node ids instead of pointers */
typedef struct {
	int32_t c_node, p_node, a_node;
	int32_t n_cand;           /* candidates that survived filterPlacements                          */
	double wuv, ratio, wnr, loglik, height, q_place, q_taxon, anno_dist, est_loglik;
	double root_loglik;       /* with fix_root_loglik: the intended root log-likelihood (== loglik then); NaN otherwise.  The
	                           * reference's constant is (cs_end - cs_start + 1) * log(sum_i pi_i e) either way           */
} hu_place_rec;

void hu_default_opts(hu_opts* o);
const char* hu_last_error(void);
/* number of usable gfx950 devices (0 => every compute entry point fails with HU_ERR_DEVICE) */
int hu_device_count(void);

/* ---- database ---------------------------------------------------------------------------
 * replaces: hmmIn >> hmm, ptu.load(ptuIn), hmm.setSequenceMode/wingRetract
 * (src/hmmufotu.cpp:457-498) */
int hu_db_create(const hu_profile_desc* prof, const hu_tree_desc* tree, const hu_model_desc* model,
		int device, hu_db** out);
/* same from the reference's on-disk formats (.hmm text, .ptu binary) */
int hu_db_load(const char* hmm_path, const char* ptu_path, int device, hu_db** out);
/* the same keeping only the messages of the CS columns [win_start, win_start + win_len) on the device (win_len 0: all) — one shard of a database
 * held as column windows, see "column-window sharding" below */
int hu_db_load_window(const char* hmm_path, const char* ptu_path, int device, int64_t win_start, int64_t win_len, hu_db** out);
/* The reference's own TEXT forms, from memory — for a caller that holds loaded reference objects: BandedHMMP7 keeps its cost
 * matrices private and has no accessors for them (src/BandedHMMP7.h:497-548), but operator<<(ostream&, const BandedHMMP7&)
 * (src/BandedHMMP7.cpp:324-378) writes exactly what hu_profile_desc wants; likewise DNASubModel::write (src/GTR.cpp:83-104 and
 * friends).  hu_profile_parse_text: first call with NULL arrays for K and L, then with EM/EI [K+1][4], T [K+1][7], p2cs [K+1].
 * hu_model_parse_text fills type, pi and par; dg_k / dg_rate are the caller's (DiscreteGammaModel::getK / rate(k)). */
int hu_profile_parse_text(const char* text, int64_t len, int32_t* K, int32_t* L, double* EM, double* EI, double* T, int32_t* p2cs);
int hu_model_parse_text(const char* text, int64_t len, hu_model_desc* out);
/* host-only parse of the two database files (no device needed): sizes with fill = 0, then the arrays a non-NULL pointer names;
 * up / down [n][cs_len][4] are the whole message set (use it on test-sized files) */
int hu_files_parse(const char* hmm_path, const char* ptu_path, int32_t* K, int32_t* L, int32_t* n_nodes, int32_t* root,
		double* EM, double* EI, double* T, int32_t* p2cs, double* entry_cost, double* exit_cost,
		int32_t* parent, double* blen, int8_t* seq, double* height, double* up, double* down, hu_model_desc* model, int fill);
/* host-only: the spectral form P(t) = U diag(exp(lam t)) U1 the kernels use for a model (U, U1 [16] row-major, lam [4]) */
int hu_model_spectral(const hu_model_desc* model, double* U, double* lam, double* U1);
void hu_db_destroy(hu_db* db);
int hu_db_info(const hu_db* db, int32_t* K, int32_t* cs_len, int32_t* n_nodes, int32_t* root, int64_t* hbm_bytes);
/* host copies of what the readers parsed (for format tests); any pointer may be NULL */
int hu_db_get_profile(const hu_db* db, double* EM, double* EI, double* T, int32_t* p2cs,
		double* entry_cost, double* exit_cost);
int hu_db_get_tree(const hu_db* db, int32_t* parent, double* blen, int8_t* seq, double* height);
int hu_db_get_model(const hu_db* db, hu_model_desc* out);
/* PTUNode::getAnno() of a node of a database loaded with hu_db_load ("" otherwise) */
const char* hu_db_get_annotation(const hu_db* db, int32_t node);
/* device-side DNASubModel::Pr(t) (src/GTR.h:116-121 and friends), for parity tests */
int hu_db_model_pr(const hu_db* db, int n, const double* t, double* P /* [n][16] row-major */);

/* ---- tree pre-evaluation (the core of hmmufotu-build) ---------------------
 * Messages of every directed edge by a post-order and a pre-order sweep, equivalent to the reference's
 * "setRoot(i); evaluate()" loop over all nodes (src/hmmufotu-build.cpp:454-459), with the discrete
 * Gamma averaging of src/PhyloTreeUnrooted.cpp:320-346; ancestral sequences by per-site argmax
 * (src/PhyloTreeUnrooted.cpp:1085-1093), node heights (:274-287).  seq [n][cs_len] holds the leaf
 * rows on entry; inner rows are filled inside the column window.  up_dev/down_dev are DEVICE buffers
 * [n][win_len][4] (log space, the .ptu convention; up[root] = root message) that can be handed to
 * hu_db_create with msgs_on_device = 1. */
int hu_tree_evaluate(int32_t n_nodes, int32_t cs_len, const int32_t* parent, const double* blen, int8_t* seq,
		const hu_model_desc* model, int device, int64_t win_start, int64_t win_len, double* up_dev, double* down_dev, double* height);

/* PTUnrooted::save (src/PhyloTreeUnrooted.cpp:537-567): writes the .ptu database file hmmufotu and hu_db_load read, from the arrays of a
 * hu_tree_desc (whole messages: win_len 0; up / down may be DEVICE buffers as hu_tree_evaluate leaves them — msgs_on_device = 1 — and are
 * then streamed to the file edge by edge).  names / annos: n C strings each (NULL: "n<i>" / "").  model_text: the model block as the
 * reference writes it (DNASubModel::write), or NULL to have it generated from `model`.  dg_alpha, dg_breaks [dg_k + 1]: what
 * DiscreteGammaModel::save stores beside the rates (the engine itself reads only the rates). */
int hu_ptu_write(const char* path, const hu_tree_desc* tree, const char* const* names, const char* const* annos, const hu_model_desc* model,
		const char* model_text, double dg_alpha, const double* dg_breaks);

/* ---- the tree of a .ptu without its messages (host only, no device): what the consumers of an assignment file need of the database —
 * hmmufotu-sum the nodes' taxon annotations (src/hmmufotu-sum.cpp:378-383), hmmufotu-jplace the topology, branch lengths and the order of
 * every node's children as PTUnrooted::load leaves it (src/hmmufotu-jplace.cpp:197, src/PhyloTreeUnrooted.cpp:1135-1157).  The 4 x csLen
 * doubles of every directed edge are read past, never kept. */
typedef struct hu_tree_info hu_tree_info;
int hu_tree_info_load(const char* ptu_path, hu_tree_info** out);
void hu_tree_info_free(hu_tree_info* t);
int hu_tree_info_get(const hu_tree_info* t, int32_t* n_nodes, int32_t* cs_len, int32_t* root, hu_model_desc* model);
/* node i: parent (-1: root), length of the branch to it, annotation distance, leaf flag; the strings live as long as t */
int hu_tree_info_node(const hu_tree_info* t, int32_t i, int32_t* parent, double* blen, double* anno_dist, int32_t* is_leaf,
		const char** name, const char** anno);
/* children of node i in the order of PTUNode::neighbors (the order their parent -> child edges stand in the file); returns the count */
int hu_tree_info_children(const hu_tree_info* t, int32_t i, const int32_t** children);

/* BandedHMMP7::buildAlignPath (src/BandedHMMP7.cpp:894-941): the CSLoc of a CSFM hit (1-based CS
 * start/end + the gapped CS string, src/CSLoc.h) and the seed's 1-based read range -> the
 * ViterbiAlignPath row {start,end,from,to,nIns,nDel} hu_batch_set_reads takes.  Host only. */
int hu_build_align_path(const hu_db* db, int cs_start, int cs_end, const char* cs, int cs_from, int cs_to, int32_t* out6);

/* ---- host seed lookup -----------------------------------------------------
 * CSFMIndex::locateOne + BandedHMMP7::buildAlignPath as alignSeq uses them (src/HmmUFOtu_main.cpp:50-84), over the same
 * text as the reference's FM-index (the gap-free leaf sequences, src/CSFMIndex.cpp:288-330), indexed here by a position
 * array sorted on the 32 symbols that follow (a depth-32 suffix array) with a 12-base directory.  The hit taken is the
 * FIRST of the suffix-ordered range (the reference's locateFirst, :92-119; its locateOne draws a random member of the same
 * range with rand(), src/CSFMIndex.cpp:139).  seed_len in 12..31 (CLI: 15..25).  Host only. */
typedef struct hu_seed_index hu_seed_index;
int hu_seed_index_create(int32_t n_nodes, int32_t cs_len, const int32_t* parent, const int8_t* seq,
		int32_t K, const int32_t* p2cs, int32_t seed_len, hu_seed_index** out);
/* The same index from the reference's own <DB>.csfm (CSFMIndex::load, src/CSFMIndex.cpp:200-230: libcds BitSequenceRRR +
 * WaveletTreeNoptrs): the sequences are recovered from the BWT, the sampled suffix array and concat2CS, and the seeds are kept in
 * the file's own suffix order, so the first hit of a seed is CSFMIndex::locateFirst's (src/CSFMIndex.cpp:92-119).
 * p2cs / K: the profile's map (hu_db_info / the .hmm), as for hu_seed_index_create. */
int hu_seed_index_load_csfm(const char* path, int32_t K, const int32_t* p2cs, int32_t seed_len, hu_seed_index** out);
void hu_seed_index_destroy(hu_seed_index* ix);
int64_t hu_seed_index_size(const hu_seed_index* ix);          /* distinct seed_len-mers */
int64_t hu_seed_index_bytes(const hu_seed_index* ix, int64_t* positions /* indexed k-mer starts; may be NULL */);   /* resident bytes */
/* every occurrence of one seed in index (suffix) order — the range locateOne draws from: sequence number (leaves in node-id
 * order), residue offset inside it, 0-based CS column of its first base.  Returns the count; at most cap entries are written */
/* CSFMIndex::locateFirst + count for one seed of the index's length: 1-based CS columns of the first hit's first and last base */
int hu_seed_index_locate_first(const hu_seed_index* ix, const char* kmer, int32_t* cs_start, int32_t* cs_end, int64_t* count);
int64_t hu_seed_index_occurrences(const hu_seed_index* ix, const char* kmer, int32_t* seq_no, int32_t* offset, int32_t* cs_col, int64_t cap);
/* Host only, no device: what std::sort (libstdc++: introsort — median-of-3 Hoare partitions down to 16 elements, heap sort past a depth of
 * 2 lg n, one final insertion sort) leaves in the first k places of n PTLocs compared on dist alone, given in node order — the computation
 * behind HU_SEED_ORDER_LIBSTDCXX, exposed for its parity test.  dist must hold no NaN (std::sort is undefined on them).  out_idx [min(k, n)]:
 * indices into dist. */
int hu_sort_prefix_libstdcxx(const double* dist, int64_t n, int64_t k, int32_t* out_idx);
/* The same on the device (k_seed_refsort, the kernel behind HU_SEED_ORDER_LIBSTDCXX), exposed for its parity test: rows x n (d, N) pairs
 * (d << 16 | N, d <= N, N >= 1; pair16 != 0: held as 16-bit pairs, d, N <= 255), one sort per row.  out_idx [rows][k]; out_cnt [rows] =
 * min(k, n), or -1 for a row the kernel left to the host path (heap-sort branch of introsort). */
int hu_sort_prefix_device(int device, const uint32_t* pairs, int rows, int64_t n, int k, int pair16, int32_t* out_idx, int32_t* out_cnt);
/* The same with the root of the test tree at place root_place (0 .. n) of node order instead of behind the last node: the kernel's level 0 is
 * the pair row WITHOUT the root's entry, read in aligned vectors that are one element off from the root on. */
int hu_sort_prefix_device_at(int device, const uint32_t* pairs, int rows, int64_t n, int k, int pair16, int64_t root_place, int32_t* out_idx, int32_t* out_cnt);
/* The device routine behind filterPlacements and the final sort (hu_kern_rank.h), exposed for its parity test: rows x n doubles (n <= HU_MAX_SEEDS),
 * order [rows][n] = for every place the index of the element that std::sort(rbegin, rend, less) — libstdc++'s introsort incl. its heap-sort
 * branch — leaves there (descending; equal keys as the library leaves them). */
int hu_sort_desc_device(int device, const double* keys, int rows, int n, int32_t* order);
/* the 5' and (GLOBAL mode) 3' seed scans for n reads -> vpaths [n][2][6] for hu_batch_set_reads */
int hu_seed_index_lookup(const hu_seed_index* ix, int n, const char* bases, const int64_t* offs, int seed_region,
		int align_mode, int32_t* vpaths);

/* The same with CSFMIndex::locateOne's hit choice (src/CSFMIndex.cpp:121-147; `hmmufotu -S <seed>`): each seed takes a pseudo-random member
 * of its hit range instead of the first.  The reference draws with rand() from ONE global stream, seeded by -S (default: the time) and
 * raced on by its OpenMP tasks, so its own runs only repeat with -p 1; here the draw is a hash of (seed, number of the read
 * in the input = first_read + its index, seed position): the same reads get the same hits whatever the batching or thread count. */
int hu_seed_index_lookup_random(const hu_seed_index* ix, int n, const char* bases, const int64_t* offs, int seed_region,
		int align_mode, uint64_t seed, int64_t first_read, int32_t* vpaths);

/* ---- batch ------------------------------------------------------------------------------ */
int hu_batch_create(hu_db* db, int max_reads, hu_batch** out);
void hu_batch_destroy(hu_batch* b);

/* reads after the host seed lookup (the boundary of SURVEY.md F7): bases are the read
 * characters (upper-case IUPAC; anything PrimarySeq::encodeAt maps to <0 marks the read
 * invalid), vpaths the ViterbiAlignPath {start,end,from,to,nIns,nDel} of the <=2 CSFM seeds
 * (src/HmmUFOtu_main.cpp:52-84, src/BandedHMMP7.cpp:894-941), rows with start==0 unused.
 * mates (already reverse-complemented, src/hmmufotu.cpp:609) may be NULL for SE. */
int hu_batch_set_reads(hu_batch* b, int n, const char* bases, const int64_t* offs, const int32_t* vpaths,
		const char* mates, const int64_t* moffs, const int32_t* mvpaths);
/* alternative entry after alignment: DigitalSeq codes [n][cs_len] and 0-based inclusive
 * regions, as getSeed/estimateSeq/placeSeq receive them (src/hmmufotu.cpp:641-645) */
int hu_batch_set_aligned(hu_batch* b, int n, const int8_t* codes, const int32_t* start, const int32_t* end);

/* alignSeq(hmm, csfm, read, ...) minus the CSFM lookup, + PE merge + DigitalSeq encoding
 * (src/HmmUFOtu_main.cpp:86-104, src/hmmufotu.cpp:621-641) */
int hu_align_batch(hu_batch* b, const hu_opts* o);
/* getSeed + truncation to max_nseed (src/HmmUFOtu_main.cpp:127-152, src/hmmufotu.cpp:645-647);
 * order is (dist, node id) — the reference's std::sort order is unspecified among ties */
int hu_seed_batch(hu_batch* b, const hu_opts* o);
/* the segment form of the seed stage (src/hmmufotu.cpp:662-665, 682, 686): the node ids are GIVEN
 * (ids [n][stride], n_seeds[r] <= HU max_nseed of them used) and only their distance is measured over the
 * batch's current regions — against dist_ids[r][s] when dist_ids is not NULL (the alt-placement PTLoc
 * takes the distance to another node than the branch it names), else against ids[r][s] */
int hu_seed_batch_given(hu_batch* b, const int32_t* n_seeds, const int32_t* ids, const int32_t* dist_ids, int stride);
/* estimateSeq over the seeds (src/HmmUFOtu_main.cpp:154-160, src/PhyloTreeUnrooted.cpp:849-877) */
int hu_estimate_batch(hu_batch* b, const hu_opts* o);
/* filterPlacements (src/HmmUFOtu_main.cpp:162-173) — host, literally std::sort */
int hu_filter_batch(hu_batch* b, const hu_opts* o);
/* placeSeq over the survivors (src/HmmUFOtu_main.cpp:175-180, src/PhyloTreeUnrooted.cpp:879-954) */
int hu_place_batch(hu_batch* b, const hu_opts* o);
/* The candidates of every read GIVEN by the caller instead of produced by hu_filter_batch — for callers that keep the reference's
 * per-stage functions (estimateSeq -> filterPlacements -> placeSeq -> calcQValues on a vector<PTPlacement>, src/HmmUFOtu_main.h:91-107)
 * and may drop, reorder or edit placements between the stages (hmmufotu_amd/csrc/hu_reference_api.hpp does).  offs [n + 1]; recs
 * [offs[n]] in the order the later stages are to see them, at most HU_MAX_SEEDS per read, none for a read that is not HU_READ_OK.
 * Read from a record: c_node (a non-root node), ratio, wnr, est_loglik; with placed != 0 also loglik, height, a_node — records as
 * hu_batch_get_candidate_places returns them — and the batch then counts as placed (hu_finish_batch may follow); with placed == 0 it
 * counts as filtered (hu_place_batch may follow).  The batch must be aligned (hu_align_batch or hu_batch_set_aligned). */
int hu_batch_set_candidates(hu_batch* b, const int64_t* offs, const hu_place_rec* recs, int placed);
/* calcQValues + final sort + bestPlace (src/HmmUFOtu_main.cpp:182-216, src/hmmufotu.cpp:725-733) */
int hu_finish_batch(hu_batch* b, const hu_opts* o);
/* the whole per-read task body for the batch (src/hmmufotu.cpp:621-733) */
int hu_assign_batch(hu_batch* b, const hu_opts* o);

/* ---- chimera check (-C; src/hmmufotu.cpp:653-691) ------------------------------------------
 * hu_chimera_opts mirrors --num-segment / --chimera-err / --chimera-lod (src/hmmufotu.cpp:145-149, 249-256,
 * 325-340): num_seg even in [2,6]; max_chimera_error > 0 (CLI default max_error / num_seg);
 * min_chimera_lod >= 0 */
typedef struct {
	int32_t num_seg;
	int32_t reserved;
	double max_chimera_error;
	double min_chimera_lod;
} hu_chimera_opts;
/* per read: the best 5' and 3' segment placements and the log-odds against each other's branch.
 * checked = 0 (and ids -1, lod NaN) for reads that are not HU_READ_OK, have no seed, or whose region is
 * shorter than num_seg columns — the reference indexes an empty vector there */
typedef struct {
	int32_t checked, is_chimera;
	int32_t seg5_start, seg5_end, seg3_start, seg3_end;   /* 0-based inclusive segment of each winner */
	int32_t n_seg5, n_seg3;                               /* pooled placements per half               */
	hu_place_rec seg5, seg3;                              /* a_node = getTaxonId()                    */
	double alt5_loglik, alt3_loglik;
	double lod;
} hu_chimera_rec;
void hu_default_chimera_opts(const hu_opts* o, hu_chimera_opts* co);
/* b: a batch that is at least seeded (hu_seed_batch done; it is left untouched and can go on to
 * hu_estimate_batch ... for the reads that are not chimeric).  work: a second batch on the same database
 * with max_reads >= b's read count; its contents are overwritten (num_seg + 2 passes of the segment
 * seed/estimate/filter/place stages run in it).  out [n] */
int hu_chimera_batch(hu_batch* b, hu_batch* work, const hu_opts* o, const hu_chimera_opts* co, hu_chimera_rec* out);

/* Kernel-selection / diagnostic knobs of a batch (which Viterbi kernel, node-ordered launches, split placement slots ...;
 * the list with defaults is HuKnobs in hmmufotu_amd/csrc/hu_engine.hip).  A batch reads its defaults from the environment
 * (HU_<NAME>) once, in hu_batch_create; this call changes one of them afterwards.  Results do not depend on any knob
 * beyond the documented tolerances — the parity tests force each alternative kernel through here.  HU_ERR_ARG: no such knob. */
int hu_batch_set_knob(hu_batch* b, const char* name, int value);

/* wait for everything queued on the batch's stream */
int hu_batch_sync(hu_batch* b);

/* ---- results (device -> caller's host buffers; any pointer may be NULL) ----------------- */
int hu_batch_get_alignments(hu_batch* b, hu_align_rec* recs, char* align /* [n][cs_len] */, char* trace, int trace_stride);
int hu_batch_get_codes(hu_batch* b, int8_t* codes /* [n][cs_len] */, int32_t* start, int32_t* end);
int hu_batch_get_pdist(hu_batch* b, int read, int32_t* d /* [n_nodes] */, int32_t* N);
/* n_seeds [n]; ids, d, N: [n][HU_MAX_SEEDS] — every row is written in full (entries past n_seeds[r]: id -1, d = N = 0) */
int hu_batch_get_seeds(hu_batch* b, int32_t* n_seeds, int32_t* ids, int32_t* d, int32_t* N);
/* ratio, wnr, loglik: [n][HU_MAX_SEEDS], NaN past the read's seed count */
int hu_batch_get_estimates(hu_batch* b, double* ratio, double* wnr, double* loglik);
/* the same with a caller-chosen row stride (>= 1): min(stride, HU_MAX_SEEDS) entries are written per read, nothing beyond;
 * for callers that size their buffers [n][max_nseed] */
int hu_batch_get_seeds_strided(hu_batch* b, int32_t* n_seeds, int32_t* ids, int32_t* d, int32_t* N, int stride);
int hu_batch_get_estimates_strided(hu_batch* b, double* ratio, double* wnr, double* loglik, int stride);
/* all candidates after placement, in filterPlacements order: offs [n+1]; arrays sized offs[n] */
int hu_batch_get_candidates(hu_batch* b, int64_t* offs, int32_t* c_node, double* ratio, double* wnr, double* est_loglik, int32_t* iters);
int hu_batch_get_placements(hu_batch* b, hu_place_rec* best /* [n] */);
/* every candidate's PTPlacement after hu_finish_batch, in filterPlacements order (offs as above) */
int hu_batch_get_candidate_places(hu_batch* b, hu_place_rec* recs /* [offs[n]] */);

/* One TSV line per read exactly as the main loop prints it (src/hmmufotu.cpp:736-739: id, description,
 * HmmAlignment operator<< src/BandedHMMP7.cpp:1215-1221, PTPlacement::write
 * src/PhyloTreeUnrooted.h:1611-1617; doubles at the default 6-significant-digit ostream precision).
 * ids/descs: n C strings; annos: taxon annotation per NODE (may be NULL -> empty column).  Reads whose
 * status is not HU_READ_OK produce no line, like the reference.  Writes into buf (capacity cap) and
 * returns the number of bytes needed (call again with a larger buffer if > cap). */
int64_t hu_batch_format_tsv(hu_batch* b, const char* const* ids, const char* const* descs, const char* const* annos,
		char* buf, int64_t cap);
/* the same with -C: which = 0 writes the assignment file's lines (HU_READ_OK reads the check did not flag), which = 1 the
 * --chimera-out lines (bad PE orientation, or flagged; placement columns of a default PTPlacement, src/hmmufotu.cpp:693-706).
 * which = 2 is the assignment file of --align-only: as 0 for a batch that is only aligned, default placement columns.
 * chimera_info != 0 inserts the --chimera-info columns before the placement (src/hmmufotu.cpp:57, 742-746).  chi [n] from
 * hu_chimera_batch (NULL: nothing was checked) */
int64_t hu_batch_format_tsv_chimera(hu_batch* b, const char* const* ids, const char* const* descs, const char* const* annos,
		const hu_chimera_rec* chi, int chimera_info, int which, char* buf, int64_t cap);
/* the same without a copy: *text points at the batch's own buffer holding the lines (valid until the next format call on this
 * batch); returns their length.  One pass — the (buf, cap) forms above format once per call, so "ask for the size, then fetch" costs
 * two.  The lines of a batch are formatted on the host thread pool of the calling thread. */
int64_t hu_batch_format_tsv_ptr(hu_batch* b, const char* const* ids, const char* const* descs, const char* const* annos,
		const hu_chimera_rec* chi, int chimera_info, int which, const char** text);
/* the bytes each read's line takes in the text of the LAST format call on this batch, newline included (0: the read has no line there): lens [n].
 * For callers that reassemble lines from several batches in read order (the column-window mode of the CLI) */
int hu_batch_tsv_line_lengths(hu_batch* b, int64_t* lens);
/* header with the --chimera-info columns (src/hmmufotu.cpp:592-594 with CHIMERA_TSV_HEADER) */
const char* hu_tsv_header_chimera(void);
/* the header line of the assignment file (src/hmmufotu.cpp:592-594) */
const char* hu_tsv_header(void);

/* ---- column-window sharding (last row) -------------------------------
 * PTUnrooted::load keeps the messages of EVERY column of every directed edge in one address space (src/PhyloTreeUnrooted.cpp:496-535); a database whose
 * messages exceed one GPU's HBM (72 bytes per node and column once packed: 28.8 MB per column at 399,999 nodes) is held as W column WINDOWS instead,
 * one hu_db per window (hu_db_load_window, or hu_tree_desc.win_*), usually one per device.  A read's placement touches only the columns of its
 * alignment region, so it runs unchanged on any window that holds the region; the seed scan runs on the node sequences, which every window holds whole.
 * Reads are ROUTED by where their seeds say they lie, before the alignment; a read that comes back HU_READ_OUT_OF_WINDOW (aligned, region known
 * exactly, not placed) is routed once more by its region.  No window exchanges anything with another: still no data-path collective.
 *   hu_windows_plan      W windows of equal width that cover [0, cs_len) and overlap their neighbours by `overlap` columns (choose it >= the widest
 *                        alignment region to be expected, so that every region lies inside at least one window)
 *   hu_route_by_seeds    per read the window to try first: the CS interval the read is expected to cover — from the profile positions of its
 *                        seed paths (vpaths [n][2][6] as for hu_batch_set_reads), extended by the bases left and right of the seeds; the window
 *                        that contains it with the widest margin, else the one that overlaps it most.  A read without a seed goes to window 0.
 *                        lens [n]: bases of each read (of the merged pair: pass the forward read's and the mate's through lens / mate_lens)
 *   hu_route_by_region   the same from exact 1-based inclusive CS regions (hu_align_rec.cs_start / cs_end); -1 when no window contains the region */
typedef struct { int64_t win_start, win_len; } hu_window;
int hu_windows_plan(int64_t cs_len, int n_win, int64_t overlap, hu_window* out /* [n_win] */);
int hu_route_by_seeds(const hu_db* db, int n_win, const hu_window* win, int n, const int32_t* lens, const int32_t* vpaths,
		const int32_t* mate_lens, const int32_t* mate_vpaths, int32_t* window_of_read /* [n] */);
int hu_route_by_region(int n_win, const hu_window* win, int n, const int32_t* cs_start, const int32_t* cs_end, int32_t* window_of_read /* [n] */);

/* ---- measurement ------------------------------------------------------------------------
 * per-kernel device time of the LAST call of each stage, measured with HIP events on the
 * batch's stream: ms[HU_T_*]; and the algorithmic work it covered */
enum { HU_T_VITERBI = 0, HU_T_ALIGN_BUILD = 1, HU_T_SEED_PDIST = 2, HU_T_SEED_TOPK = 3,
	HU_T_ESTIMATE = 4, HU_T_PLACE = 5, HU_T_COUNT = 8 };
int hu_batch_timings(hu_batch* b, float* ms /* [HU_T_COUNT] */);
/* enable/disable event recording around kernels (off by default: zero overhead) */
int hu_batch_profile(hu_batch* b, int enable);
/* host wall-clock (ms) of the last hu_assign_batch: align | seed+estimate+filter | place | finish */
int hu_batch_wall(hu_batch* b, double* ms4);
/* the last hu_seed_batch under HU_SEED_ORDER_LIBSTDCXX: reads the device sort (k_seed_refsort) handed to the host restatement (heap-sort branch, table
 * overflow; the NaN-distance reads of a database with partial sequences are counted too), and whether the WHOLE batch took the host path (a tree beyond
 * the kernel's LDS index, ~9 M nodes: said once on stderr) — so that a rate quoted for the mode is never silently a host-path rate */
int hu_batch_refsort_stats(hu_batch* b, int32_t* left_to_host, int32_t* whole_batch_on_host);

#ifdef __cplusplus
}
#endif
#endif /* HMMUFOTU_AMD_H_ */
