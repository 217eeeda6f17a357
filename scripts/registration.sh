#!/bin/bash
## Drop-in for the reference's scripts/registration.sh (python PointCloud/mlp_reg.py --robot wx200_5):
## same flags, same data/raw -> data/part layout, run from a directory holding parameters.json.
python -m autourdf_amd.mlp_reg --robot wx200_5 "$@"
