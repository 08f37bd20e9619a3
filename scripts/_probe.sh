for wl in city_swin_l_k10_4x1024x2048 kitti_depth_k20_16x352x1216 bev_fusion_k3_8x200x200; do
  echo "== $wl"; timeout 300 python bench.py --workload $wl --steps 3 --warmup 1 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], 'img/s', d['ms_per_step'], 'ms/step', r['avg_launch_ms'], 'ms/launch', r['frac'], r['loop_tflops'])"
done
