"""The GPU suite with its test FILES in a seeded random order (pytest runs files in command-line order; no plugin needed).
Round 5's one "flaky" failure was an order dependence (profiles/r06_zero_copy_flake.txt): the suite must pass in any file order.
  python tools/gpu_suite_shuffled.py <seed> [extra pytest args]"""
import glob
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "test_*.py")))
    random.Random(seed).shuffle(files)
    print("seed %d order: %s" % (seed, " ".join(os.path.basename(f)[5:-3] for f in files)), flush=True)
    sys.exit(subprocess.call([sys.executable, "-m", "pytest"] + files + ["-m", "gpu", "-q", "--tb=short", "-p", "no:cacheprovider"] + sys.argv[2:], cwd=ROOT))


if __name__ == "__main__":
    main()
