#!/bin/bash
# Do non-temporal record stores (the default; -DH263MI_NO_NT_RECORDS = plain stores) pay in the host parser when 16 threads write their records into a
# 16.7 MB ring (the shape of the batch entry's pinned staging)?  tools/parse_scaling.cpp, both builds, one realistic 1080p
# P picture and one key frame.  usage (GPU box, repo root): bash tools/probes/nt_records.sh > gpurun_out/nt_records.txt
set -u
R=$PWD
python3 - <<'PY'
import sys
sys.path[:0] = ['.', 'tests', 'h263-rs_amd']
import recgen, sorenson_enc as enc
from test_bitstream_e2e import make_codable
W, H = 1920, 1080
mbs, co = recgen.realistic_inter_picture(W, H, 7001)
open('/tmp/real_P.bin', 'wb').write(enc.encode_picture(W, H, 1, 10, make_codable(mbs, 10, 1, 1), co))
mbs, co = recgen.realistic_intra_picture(W, H, 300)
open('/tmp/real_I.bin', 'wb').write(enc.encode_picture(W, H, 0, 10, make_codable(mbs, 10, 0, 0), co))
PY
g++ -O3 -std=c++17 -pthread -Iinclude -DH263MI_NO_NT_RECORDS -o /tmp/ps_plain tools/parse_scaling.cpp h263-rs_amd/host/bitstream.cpp
g++ -O3 -std=c++17 -pthread -Iinclude -o /tmp/ps_nt tools/parse_scaling.cpp h263-rs_amd/host/bitstream.cpp
for rep in 1 2; do
  for f in real_P real_I; do
    echo "== $f, plain stores (run $rep)"; /tmp/ps_plain /tmp/$f.bin 300 | grep " 1 threads\|16 threads"
    echo "== $f, non-temporal record stores (run $rep)"; /tmp/ps_nt /tmp/$f.bin 300 | grep " 1 threads\|16 threads"
  done
done
