"""Stand-in for the nine `torch_geometric` names the reference imports.

Used ONLY by tests/golden/make_goldens.py, in the build container, to import the reference's
unmodified networks.py / data.py (torch_geometric is not installed there and there is no
network).  Written from PyG's published definitions of these ops; it is
not PyG and never ships in the product path.
"""
__version__ = "0.0-standin"
