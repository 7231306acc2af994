#!/bin/bash
# round 3, session O: tile merge spread over several waves per sequence (parity, rate)
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests/test_gpu_long_sequences.py tests/test_gpu_parity.py::test_long_reads_and_ragged_batches tests/test_gpu_parity.py::test_cli_contig_queries_multiline_fasta -m gpu -x -q 2>&1 | tail -3
python scripts/time_contigs.py 400000 8 2>&1 | tail -3
python scripts/time_contigs.py 400000 1 2>&1 | tail -3
python scripts/time_contigs.py 5000 2000 2>&1 | tail -3
python scripts/time_contigs.py 50000 200 2>&1 | tail -3
