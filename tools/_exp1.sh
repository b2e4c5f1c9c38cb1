for cfg in "A::" "B:S2SVC_NO_BRANCH=1:" "C::--side-streams 0" "D:S2SVC_NO_BRANCH=1:--side-streams 0" "E::--side-streams 2" "F::--side-streams 4"; do
  n=${cfg%%:*}; rest=${cfg#*:}; envv=${rest%%:*}; fl=${rest#*:}
  echo "== $n env=[$envv] flags=[$fl]"
  env $envv python bench.py --workload aasvc --no-cpu-baseline --steps 50 --warmup 5 $fl 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
