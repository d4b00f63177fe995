#!/bin/bash
# Developer tool (GPU box): tests/native/td_linearize_gpu with large crowds of caller threads, each run verified
# against the reference (tests/td_scenarios.py:verify_linearizable). Usage: bash tests/tools/soak_linearize.sh
cd /root/repo
for cfg in "2000 64 3000 0 21" "40 1000 60 1 22" "200 300 300 2 23" "2000 128 1500 0 24"; do
  set -- $cfg
  echo "== td_linearize_gpu servants=$1 threads=$2 calls=$3 cap=$4 seed=$5"
  timeout 600 tests/native/td_linearize_gpu /tmp/lin_$5.json $1 $2 $3 $4 $5 || echo "FAILED rc=$?"
  timeout 900 python - <<EOF
import sys, json, time
sys.path.insert(0, '/root/repo')
from tests import td_scenarios as S
j = json.load(open('/tmp/lin_$5.json'))
threads = [{"ip": t["ip"], "ops": [(o[0], tuple(o[1:4]), o[4], o[5]) if o[0] == "wait" else (o[0], o[1], o[2], o[3]) for o in t["ops"]]} for t in j["threads"]]
t0 = time.time()
n = S.verify_linearizable(j["log"], j["dump"], threads)
print("  verified", n, "records in %.1f s; requests per device turn %.2f; retried attempts %d" % (time.time() - t0, j["requests_per_device_turn"], sum(1 for e in j["log"] if e["op"] == "wait" and e["try"] > 1)))
EOF
done
