"""Alias of freepose_amd.scripts.merge_features (reference module path: scripts/merge_features.py)."""
from freepose_amd.scripts.merge_features import *  # noqa: F401,F403
from freepose_amd.scripts.merge_features import main

if __name__ == "__main__":
    main()
