#!/bin/bash
echo "== slp on"; HQQ_AMD_LIB=$PWD/tools/r6/libhqq_hip_slp.so python tools/r6/dbg_bits4.py 2>/dev/null | tail -4
echo "== cur (no slp)"; python tools/r6/dbg_bits4.py 2>/dev/null | tail -4
