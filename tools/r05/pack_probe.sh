#!/bin/bash
# jobs of one or two strips sharing four-wave workgroups (CSADP_BITS_PACK=1): config 4 at four words per lane (two strips per job), and the default
cd ${GRAFT_REPO_ROOT:-.}
run() {
  python bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-extra-legs "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$TAG: %.1f TCUPS  %.3f ms/step  words %s passes/launch %s streams %s verified %s' % (d['value']/1e3, d['ms_per_step'], d['config'].get('words_per_lane'), d['config'].get('passes_per_launch'), d['config'].get('launches_in_flight'), d.get('verified')))"
}
for rep in 1 2; do
TAG="default (two words)"; run
TAG="four words, packed"; CSADP_BITS_PACK=1 CSADP_BITS_WORDS=4 run
TAG="four words, packed, streams 2 group 4"; CSADP_BITS_PACK=1 CSADP_BITS_WORDS=4 CSADP_BITS_STREAMS=2 CSADP_BITS_GROUP=4 run
TAG="four words, packed, streams 2 group 8"; CSADP_BITS_PACK=1 CSADP_BITS_WORDS=4 CSADP_BITS_STREAMS=2 CSADP_BITS_GROUP=8 run
TAG="four words, packed, streams 4 group 4"; CSADP_BITS_PACK=1 CSADP_BITS_WORDS=4 CSADP_BITS_STREAMS=4 CSADP_BITS_GROUP=4 run
TAG="four words, packed, streams 3 group 4"; CSADP_BITS_PACK=1 CSADP_BITS_WORDS=4 CSADP_BITS_STREAMS=3 CSADP_BITS_GROUP=4 run
TAG="four words, unpacked"; CSADP_BITS_WORDS=4 run
TAG="8 kbp x 256, two words"; run --len 8192 --pairs 256
TAG="8 kbp x 256, two words, packed"; CSADP_BITS_PACK=1 run --len 8192 --pairs 256
TAG="4 kbp x 512"; run --len 4096 --pairs 512
TAG="4 kbp x 512, packed"; CSADP_BITS_PACK=1 run --len 4096 --pairs 512
done
