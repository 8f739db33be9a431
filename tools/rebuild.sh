#!/bin/bash
# rebuild the in-tree artefacts (libngp_hip.so, pyngp, the CLI, the oracle) from any working directory
cd "$(dirname "$0")/.." && python3 -c "
import importlib
b = importlib.import_module('surface-irradiance-estimation-from-neural-radiance-fields_amd.build')
b.build(); b.build_pyngp(); b.build_main()" && make -s -C oracle OUT=$PWD/oracle/_build all
