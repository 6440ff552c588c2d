#!/usr/bin/env python3
"""Same name and flags as the reference's src/py_scripts/fc_graphs_to_h_tigs.py (-> falcon_unzip.graphs_to_h_tigs.main)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from falcon_unzip_amd.graphs_to_h_tigs import main

if __name__ == "__main__":
    main(sys.argv)
