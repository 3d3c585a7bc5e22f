"""One-screen summary of a bench.py line: python tools/sum_bench.py <file with the JSON line>"""
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print(d["value"], d["roofline"]["frac"], d["roofline"]["peak_measured"])
for k in ("e2e_bitstream","e2e_bitstream_realistic"):
    print(k, d["extra"][k]["pictures_per_s"], d["extra"][k]["parity_vs_oracle"])
print("denseI", d["extra"]["config2_dense_iframe"]["pipeline_frac"])
