cd $GRAFT_REPO_ROOT
for cfg in "1 2 -1" "1 4 -1" "1 4 30" "1 8 0" "1 8 32" "1 6 20"; do set -- $cfg
  CSADP_BITS_LDS_PAD=$3 CSADP_BITS_STREAMS=$1 CSADP_BITS_GROUP=$2 python bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms']
print('streams $1 group $2 pad $3: %7.0f GCUPS %.3f ms/step  fill alone %.3f ms for %d passes = %.3f per pass; tb %.3f' % (d['value'], d['ms_per_step'], k['fill_launch_alone'], k['passes_in_that_launch'], k['fill_launch_alone']/k['passes_in_that_launch'], k['traceback_and_expand_alone']))"
done
