"""`python src/retrievers/bm25.py ...` -- the reference's lexical-retriever CLI path (scripts/run_bm25.sh), served by fusion_amd
(see fusion_amd/retrievers/bm25.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion_amd.retrievers.bm25 import BM25, TFIDF, AtireBM25, build_parser, main  # noqa: E402,F401

if __name__ == "__main__":
    args, _ = build_parser().parse_known_args()
    main(args)
