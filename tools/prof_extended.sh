#!/bin/bash
# per-kernel table of the extended-stage operators at T = 2^22 (tools/time_extended.py under rocprofv3)
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/p_ext
timeout 300 rocprofv3 --kernel-trace -d /tmp/p_ext -o e -- python /root/repo/tools/time_extended.py ${1:-22} > /tmp/ext.txt 2>&1
f=$(find /tmp/p_ext -name "*.db" | head -1); python /root/repo/profiles/summarize_rocprof.py "$f" | head -${2:-40} | cut -c1-165
