cd $GRAFT_REPO_ROOT
cp csa_amd/libcsadp.so /tmp/base.so; cp build/libcsadp_tbtimers.so csa_amd/libcsadp.so
python tools/single_probe.py 16384 2>&1 | grep -E "timers|fill" | sort | uniq -c | sort -rn | head -6
cp /tmp/base.so csa_amd/libcsadp.so
