"""Synthetic workloads of SURVEY.md section 8d (genomes, sketch tables, reads) for tests, bench.py and
tools/ -- data generators, not part of the product (mash_amd/ holds only what the path needs)."""
