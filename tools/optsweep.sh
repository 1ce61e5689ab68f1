# bench.py kernel times under the tile-kernel options (kc, item size, XCD swizzle, item order): all within noise since round 3
for o in "" "kc=32" "ls_item_chunks=64" "ls_item_chunks=128" "ls_item_chunks=32" "kc=32,ls_item_chunks=64" "xcd_swizzle=0" "ls_sort_items=0"; do
  echo "OPTS=$o"; DSH_BENCH_OPTS="$o" python bench.py --steps 10 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(l['ms_per_step'],3), l['roofline']['step']['ms'], l['roofline']['valu_int']['cycles_per_and_bcnt_pair'])"
done
