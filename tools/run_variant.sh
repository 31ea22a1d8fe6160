#!/bin/bash
# rebuild libmistitch.so on the GPU box with extra flags ($1), then run the rest of the command line
R=$GRAFT_REPO_ROOT
flags=$1; shift
cd $R/image_stitching_amd/csrc && touch *.hip && make -s CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-unused-result -Wno-bitwise-instead-of-logical $flags" > $R/gpurun_out/variant_build.log 2>&1
cd $R && "$@"
