#!/bin/bash
# Runs ON THE GPU BOX: same-box A/B of bench.py between library variants (LMH_VARIANT names; "" = the shipped library).
# usage: bash scripts/ab.sh OUTDIR "variantA variantB ..." [bench args]      (the shipped library is always measured first and last)
O=$1; shift
VARS=$1; shift
mkdir -p $O
for v in "" $VARS ""; do
  n=${v:-shipped}
  LMH_VARIANT=$v timeout -k 10 300 python bench.py --no-cpu-baseline --steps 4 --warmup 1 "$@" > $O/ab_$n.json 2> $O/ab_$n.err || echo "variant $n failed"
  python - "$O/ab_$n.json" "$n" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:>12}: {d['value']/1e6:8.3f} M ticks/s  kernel {d['roofline']['kernel_ms']:9.2f} ms  flagged {d['instances_flagged']}")
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
done
