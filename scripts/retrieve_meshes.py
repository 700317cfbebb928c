"""Alias of freepose_amd.scripts.retrieve_meshes (the retrieval half of the reference's scripts/extract_proposals_ground[_video].py)."""
from freepose_amd.scripts.retrieve_meshes import *  # noqa: F401,F403
from freepose_amd.scripts.retrieve_meshes import run

if __name__ == "__main__":
    run()
