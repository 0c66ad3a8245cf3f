#!/bin/bash
# profiles/variants.sh -- build variants of the library with other compile-time shapes of the
# general instance into rnamotif_amd/csrc/build_var/NAME/librnamotif_amd.so (selected at run time with
# RNAMOTIF_AMD_LIB); what profiles/matrix_r2.sh compares.  Not part of the product build.
set -e
cd "$(dirname "$0")/../rnamotif_amd/csrc"
make -s all
build() {	# name, flags
	mkdir -p build_var/$1
	/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -I../../include -I. --offload-arch=gfx950 $2 -c rm_scan_hip.hip -o build_var/$1/rm_scan_hip.o
	/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build_var/$1/librnamotif_amd.so $(ls build/*.o | grep -v "rm_scan_hip\|rm_main\|rm_pack_main") build_var/$1/rm_scan_hip.o
}
build pbinl "-DPASS_B_ATTR=__attribute__((always_inline))" &
build pbinl3 "-DPASS_B_ATTR=__attribute__((always_inline)) -DGENERAL_WAVES_PER_SIMD=3" &
build pbnoi3 "-DGENERAL_WAVES_PER_SIMD=3" &
wait
ls -la build_var/*/librnamotif_amd.so
