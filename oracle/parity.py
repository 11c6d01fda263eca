"""ORACLE — TEST INFRASTRUCTURE ONLY.  Classification of differences between the engine's candidate order /
final pick and the oracle's.  Imported by tests/ and by bench.py's cpu_baseline leg, never by hmmufotu_amd/.

Background: the reference ranks a read's candidates by their ESTIMATED
log-likelihood (filterPlacements, src/HmmUFOtu_main.cpp:162-173); the placed logliks all tie (F4), so the final
pick is whatever candidate sits at a fixed position of that order after std::sort's tie permutation.  Bit-exact
ids therefore need the same filterPlacements order.  One class of difference is NOT a defect of either side:

  attaching the read AT A TREE NODE X can be written on every branch incident to X — ratio 0 on branch X->parent(X)
  (estimateSeq gives ratio = cDist/(cDist+pDist) = 0 when the read is identical to X in its region) and ratio 1 on
  each branch child->X (pDist = 0).  These are the same tree with the same branch lengths, so their estimated
  log-likelihoods are equal in exact arithmetic and differ only by rounding (log-space libm arithmetic in the
  reference / oracle, linear-space eigenbasis arithmetic on the device).  Which of them sorts first is rounding
  noise in the reference itself.

A swap is EXPLAINED iff both candidates attach at the same node X (estimated ratio exactly 0 or 1 on branches
incident to X) and their oracle estimated logliks agree to NEAR_TIE relative.  Everything else is UNEXPLAINED and
must be zero.
"""
import numpy as np

NEAR_TIE = 1e-9


def attach_node(node, ratio0, parent):
    """the tree node a candidate attaches at, or None when it attaches inside its branch"""
    if ratio0 == 0.0:
        return int(node)
    if ratio0 == 1.0:
        return int(parent[int(node)])
    return None


def explained_swap(a, b, est, ratio0, parent):
    """a, b: c_node ids of two candidates of one read; est / ratio0: dicts node -> oracle value"""
    if a not in est or b not in est:
        return False
    ea, eb = est[a], est[b]
    if not (abs(ea - eb) <= NEAR_TIE * max(abs(ea), abs(eb))):
        return False
    xa, xb = attach_node(a, ratio0[a], parent), attach_node(b, ratio0[b], parent)
    return xa is not None and xa == xb


def classify_read(o_nodes, o_est, o_ratio0, g_nodes, parent, pos=None):
    """o_*: oracle candidates in filterPlacements order; g_nodes: the engine's, same stage; pos: position of the final
    pick in that order (None: not checked).  Returns a dict of counts for this read:
      swaps_explained / swaps_unexplained : positions holding another node
      set_differs                          : 1 when the candidate SETS differ (always unexplained)
      best_differs / best_unexplained      : the node at `pos` differs / differs without an explanation
    """
    o_nodes = [int(x) for x in o_nodes]; g_nodes = [int(x) for x in g_nodes]
    out = dict(swaps_explained=0, swaps_unexplained=0, set_differs=0, best_differs=0, best_unexplained=0, detail=[])
    if sorted(o_nodes) != sorted(g_nodes):
        out["set_differs"] = 1
        out["detail"].append(("set", o_nodes, g_nodes))
        if pos is not None and (pos >= len(g_nodes) or pos >= len(o_nodes) or g_nodes[pos] != o_nodes[pos]):
            out["best_differs"] = out["best_unexplained"] = 1
        return out
    est = {n: float(e) for n, e in zip(o_nodes, o_est)}
    rat = {n: float(r) for n, r in zip(o_nodes, o_ratio0)}
    for i, (a, b) in enumerate(zip(o_nodes, g_nodes)):
        if a == b:
            continue
        ok = explained_swap(a, b, est, rat, parent)
        out["swaps_explained" if ok else "swaps_unexplained"] += 1
        if not ok:
            out["detail"].append(("swap", i, a, b, est[a], est[b], rat[a], rat[b]))
        if pos is not None and i == pos:
            out["best_differs"] = 1
            out["best_unexplained"] = 0 if ok else 1
    return out


def summarize(per_read):
    tot = dict(reads=len(per_read), swaps_explained=0, swaps_unexplained=0, set_differs=0, best_differs=0, best_unexplained=0)
    for r in per_read:
        for k in tot:
            if k != "reads":
                tot[k] += r[k]
    return tot
