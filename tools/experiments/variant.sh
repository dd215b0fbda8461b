#!/bin/bash
# usage: variant.sh <name> <file.hip> <extra hipcc flags...>  -> scratch/lib_<name>.so
name=$1; f=$2; shift 2
cd /root/repo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast "$@" -c nabu_amd/csrc/$f -o scratch/var_$name.o || exit 1
objs=$(ls nabu_amd/build/*.o | grep -v "/${f%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/lib_$name.so $objs scratch/var_$name.o
