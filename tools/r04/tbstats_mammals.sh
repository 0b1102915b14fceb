cd $GRAFT_REPO_ROOT; cp csa_amd/libcsadp.so /tmp/base.so; cp build/libcsadp_tbstats.so csa_amd/libcsadp.so
timeout -k 10 120 python tools/msa_probe.py Mammals 2>&1 | grep "^resolve" | sort | uniq -c | sort -k1 -n | tail -30
cp /tmp/base.so csa_amd/libcsadp.so
