"""Alias of freepose_amd.scripts.merge_results (reference module path: scripts/merge_results.py)."""
from freepose_amd.scripts.merge_results import *  # noqa: F401,F403
from freepose_amd.scripts.merge_results import main

if __name__ == "__main__":
    main()
