"""Host-side file formats either side of the hot path: FASTA / GFF3 readers, a BGZF/BAM writer
(used to make synthetic inputs; the BAM *reader* is native, csrc/bam_reader.cpp)."""
