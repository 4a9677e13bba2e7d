#!/bin/bash
# scratch: build with -DEXP=n on the GPU box and run a command
e=$1; shift
touch slimm_amd/csrc/*.hip
make -C slimm_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC -DEXP=$e" 2>&1 | grep -E "error" 
"$@"
