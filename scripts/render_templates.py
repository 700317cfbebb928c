"""Alias of freepose_amd.scripts.render_templates (reference module path: scripts/render_templates.py)."""
from freepose_amd.scripts.render_templates import *  # noqa: F401,F403
from freepose_amd.scripts.render_templates import run

if __name__ == "__main__":
    run()
