"""`python -m scripts.<name>` entry points with the reference's module names; implementations live in
freepose_amd/scripts/."""
