#!/usr/bin/env python
"""Launcher with the reference driver's file name: `python train_mm_vi_model1.py -data ... -gpuid 0` runs
variational_mmt_amd/train_mm_vi_model1.py (the MI355X build's driver; same flags as the reference's script of this name)."""
import sys

from variational_mmt_amd.train_mm_vi_model1 import main

if __name__ == "__main__":
    main(sys.argv[1:])
