#!/usr/bin/env python3
"""Same name and flags as the reference's src/py_scripts/fc_select_reads_from_bam.py (-> falcon_unzip.select_reads_from_bam.main)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from falcon_unzip_amd.select_reads_from_bam import main

if __name__ == "__main__":
    main(sys.argv)
