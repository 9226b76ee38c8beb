cd $GRAFT_REPO_ROOT
python3 -c "
from rgbd_pose_estimation_amd import build as B
print(B.build_stamps(1))" 2>/dev/null | tail -1
so=rgbd_pose_estimation_amd/lib/librgbdpose_hip_stamps1.so
for blk in 256 512; do for nk in "307200 0" "1000000 1"; do RPE_BLOCK=$blk RPE_STEADY=12 RPE_LIBRARY=$so python3 scripts/tail_timeline.py --worker stamps $nk 60 2>/dev/null | python3 -c "
import sys, json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
keep=['G','start_last','body_done_med','body_done_max','wave_reduced_med','wave_reduced_max','wg_barrier1_max','granules_stored_med','granules_stored_max','run_read_max','run_record_sent_max','collecting_workgroups']
print('block $blk', j['n'], j['kind'], {k: round(j[k],2) for k in keep if k in j})"; done; done
