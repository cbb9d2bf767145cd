"""Sum of numer / sum of denom over the WHOLE lower triangle of the default synthetic tables
(workloads.synth_torch.clustered_sketch_table, seed 0, clusters = n / 100), keyed by (n, s).

One place: bench.py asserts its timed output against them, and the tests that establish them
(tests/test_gpu_parity.py::test_c3_full_size_triangle / test_c5_full_size_triangle) assert the same
sums after checking every within-cluster pair and a million cross-cluster pairs of that very output
against the reference's own compareSketches (oracle/_ref) and the sums of two independent engines."""
C3_CHECKSUM = {
    (100_000, 1000): (2122078313, 4999950000000),      # BASELINE config 3
    (100_000, 10000): (21217550236, 49999500000000),   # BASELINE config 5 at config-3 scale
}
