#!/usr/bin/env python3
"""cProfile of the host side of one product run (batch._one_tile on a coherent scene, single thread of tiles): where the
interpreter's own time goes between the native calls.  Prints the top functions by own time."""
import cProfile
import os
import pstats
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import make_synthetic_hls as synth_hls          # noqa: E402
from proteus_amd import batch, dswx_hls as D    # noqa: E402
import logging                                  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    logging.getLogger('dswx_hls').setLevel(logging.WARNING)
    with tempfile.TemporaryDirectory() as d:
        rc = synth_hls.make(d, scene=True)[0]
        for _ in range(2):
            assert batch._one_tile(D, 0, rc, False)['ok']
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(reps):
            batch._one_tile(D, 0, rc, False)
        pr.disable()
        st = pstats.Stats(pr)
        st.sort_stats('tottime').print_stats(35)


if __name__ == '__main__':
    main()
