"""`python src/retrievers/hybrid.py ...` -- the reference's CLI path, served by fusion_amd (see fusion_amd/retrievers/hybrid.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion_amd.retrievers.hybrid import Aggregator, Ranker, build_parser, main, run_evaluation  # noqa: E402,F401

if __name__ == "__main__":
    args, _ = build_parser().parse_known_args()
    main(args)
