#!/usr/bin/env python
# same trampoline as the reference's src/py_scripts/fc_rr_hctg_track.py, importing the MI355X engine
from falcon_unzip_amd.rr_hctg_track import main
import sys
if __name__ == "__main__":
    main(sys.argv)
