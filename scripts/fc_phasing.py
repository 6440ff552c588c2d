#!/usr/bin/env python
# same trampoline as the reference's src/py_scripts/fc_phasing.py:1-4
import sys
from falcon_unzip_amd.phasing import main
main(sys.argv)
