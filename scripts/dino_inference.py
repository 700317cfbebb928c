"""Alias of freepose_amd.scripts.dino_inference (reference module path: scripts/dino_inference.py)."""
from freepose_amd.scripts.dino_inference import *  # noqa: F401,F403
from freepose_amd.scripts.dino_inference import run

if __name__ == "__main__":
    run()
