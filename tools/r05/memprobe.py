import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import torch, bench, csa_amd
from csa_amd.synth import config4_tasks
torch.cuda.set_device(0)
csa_amd.init(device=0)
def free(tag): f, t = torch.cuda.mem_get_info(); print("%-40s free %.1f GiB" % (tag, f / 2**30), flush=True)
free("start")
tasks = config4_tasks(0, 128)
batch = csa_amd.PairBatch(tasks); batch.sync(); free("main batch created")
for _ in range(40): batch.run()
batch.sync(); free("main batch ran")
big = tasks + config4_tasks(256, 384)
bench.streaming_leg(csa_amd, big, batches=8); free("after streaming")
bench.one_shot_leg(csa_amd, tasks); free("after one_shot")
bench.profile_path_leg(csa_amd); free("after profile_path")
bench.single_matrix_leg(csa_amd); free("after single_matrix")
bench.profile_batch_leg(csa_amd); free("after profile_batch")
bench.real_sets_leg(csa_amd); free("after real_sets")
try:
    r = bench.config5_leg(csa_amd); print("config5", r["gcups"]); free("after config5")
except Exception as e:
    print("config 5:", e); free("after the failed create")
