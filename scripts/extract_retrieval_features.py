"""Alias of freepose_amd.scripts.extract_retrieval_features (reference module path: scripts/extract_retrieval_features.py)."""
from freepose_amd.scripts.extract_retrieval_features import *  # noqa: F401,F403
from freepose_amd.scripts.extract_retrieval_features import main

if __name__ == "__main__":
    main()
