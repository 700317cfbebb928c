"""Alias of freepose_amd.scripts.dino_inference_video (reference module path: scripts/dino_inference_video.py)."""
from freepose_amd.scripts.dino_inference_video import *  # noqa: F401,F403
from freepose_amd.scripts.dino_inference_video import build_parser, main, run

if __name__ == "__main__":
    run()
