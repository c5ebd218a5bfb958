#!/usr/bin/env python3
"""Which of tm_benchmark's fourteen graphs (benchmark/tm_benchmark.cc:250-289) does hip_split_graph give to device "HIP" whole, in
fp32, and which operators keep the rest on the CPU device?  Runs without a GPU (the split happens before the device is touched);
prints the markdown table of INTEGRATION.md section G.  Needs oracle/_ref (the reference library + its model files)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from oracle import ref_capi
    import test_reference_benchmark_files as t
    print("| model file | compute nodes | HIP subgraphs | nodes on HIP | operators left to the CPU device |")
    print("|---|---|---|---|---|")
    for f, n, nh, on, cpu in t.split_table(ref_capi):
        print("| `%s_benchmark.tmfile` | %d | %d | %d | %s |" % (f, n, nh, on, ", ".join(cpu) if cpu else "— (one HIP subgraph)"))


if __name__ == "__main__":
    main()
