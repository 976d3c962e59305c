#!/usr/bin/env python3
"""`dswx_compare.py file1 file2` (PROTEUS bin/dswx_compare.py:32-41)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proteus_amd.dswx_hls import compare_dswx_hls_products   # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser(description='Compare two DSWx-HLS products')
    ap.add_argument('input_file', type=str, nargs=2, help='Input images')
    args = ap.parse_args(argv)
    return 0 if compare_dswx_hls_products(args.input_file[0], args.input_file[1]) else 1


if __name__ == '__main__':
    sys.exit(main())
