"""Runs the CPU oracle (oracle/) on the seeded inputs of tests/oracle_fixtures.py and writes
tests/golden/oracle_*.npz.  Build container only (about half an hour on 8 cores):

    python tests/golden/make_oracle_fixtures.py [name ...]        # no names: every fixture that is missing
    python tests/golden/make_oracle_fixtures.py --all             # regenerate all

The files are outputs of the builder's oracle, not of the reference (which cannot be built here)."""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from tests import oracle_fixtures as F  # noqa: E402


def main():
    args = sys.argv[1:]
    names = list(F.SPECS) if (not args or args == ["--all"]) else args
    threads = max(1, len(os.sched_getaffinity(0)))
    for name in names:
        if not args and os.path.exists(F.path(name)):
            continue
        t0 = time.time()
        out = F.generate(name, threads)
        f = out["flags"]
        print(f"{name}: {len(f)} configurations, env/self/limit {[(f & b > 0).mean().round(3) for b in (1, 2, 4)]}, "
              f"uncertified candidates {len(out.get('adj_idx', []))}, {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
