#!/usr/bin/env python
# same trampoline as the reference's src/py_scripts/fc_phasing_readmap.py:1-4
import sys
from falcon_unzip_amd.phasing_readmap import main
main(sys.argv)
