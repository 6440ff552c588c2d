#!/usr/bin/env python
# the consensus role of the reference's quiver task (falcon_unzip/run_quiver.py:82-97) on the MI355X engine: fzp_polish_tigs
from falcon_unzip_amd.polish_tigs import main
import sys
if __name__ == "__main__":
    main(sys.argv)
