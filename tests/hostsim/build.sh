#!/bin/bash
# tests/hostsim/build.sh -- TEST INFRASTRUCTURE: build tests/_build/hostsim_check (the device search
# state machines compiled for the host, next to the oracle).  tests/test_hostsim.py does the same.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
H=$R/rnamotif_amd/csrc
mkdir -p $R/tests/_build
make -s -C $R/oracle liboracle.so rnamotif_oracle > /dev/null
g++ -O2 -g -std=c++17 -pthread -I$R/include -I$H -I$R/oracle -o $R/tests/_build/hostsim_check $R/tests/hostsim/hostsim_check.cpp \
	$H/rm_regex.cpp $H/rm_compile.cpp $H/rm_parse.cpp $H/rm_score.cpp $H/rm_efndata.cpp $H/rm_efn2data.cpp $H/rm_fasta.cpp \
	$H/rm_driver.cpp $H/rm_cli.cpp $H/rm_dump.cpp $H/rm_pack.cpp $H/rm_stream.cpp $H/rm_dev_program.cpp \
	$R/oracle/rm_oracle_scan.o $R/oracle/rm_oracle_efn.o $R/oracle/rm_oracle_efn2.o -lm
