// Test infrastructure: prints what the product's FASTA / FASTQ reader (hmmufotu_amd/csrc/hu_reads_io.h) makes of a file, one record per
// line as id \x1f desc \x1f seq, for tests/test_ref_seq.py to compare with the reference's own SeqIO.  Built with the sanitizers there.
// usage: reads_driver <fasta|fastq> <file>   |   reads_driver revcom <sequence>
#include <cctype>
#include <iostream>
#include "hu_reads_io.h"
int main(int argc, char** argv) {
	if(argc != 3) return 2;
	if(std::string(argv[1]) == "revcom") { std::cout << revcom(argv[2]) << '\n'; return 0; }      /* the mate's reverse complement (PrimarySeq::revcom) */
	LineIn in;
	if(!in.open(argv[2])) return 3;
	const bool fastq = std::string(argv[1]) == "fastq";
	Read r;
	while(next_read(in, fastq, r)) std::cout << r.id << '\x1f' << r.desc << '\x1f' << r.seq << '\n';
	return 0;
}
