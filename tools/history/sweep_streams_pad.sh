cd $GRAFT_REPO_ROOT
for rep in 1 2; do for cfg in "2 2 -1" "3 2 9" "4 2 5" "4 2 0" "3 2 -1"; do for run in "20 5" "48 8"; do set -- $cfg $run
  CSADP_BITS_LDS_PAD=$3 CSADP_BITS_STREAMS=$1 CSADP_BITS_GROUP=$2 python bench.py --steps $4 --warmup $5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('rep $rep streams $1 group $2 pad $3 steps $4: %7.0f GCUPS %.3f ms/step  fill alone %.3f (%d passes) tb %.3f verified %s' % (d['value'], d['ms_per_step'], k['fill_launch_alone'], k['passes_in_that_launch'], k['traceback_and_expand_alone'], d.get('verified')))"
done; done; done
