#!/usr/bin/env python
# same trampoline as the reference's src/py_scripts/fc_ovlp_filter_with_phase.py, importing the MI355X engine
from falcon_unzip_amd.ovlp_filter_with_phase import main
import sys
if __name__ == "__main__":
    main(sys.argv)
