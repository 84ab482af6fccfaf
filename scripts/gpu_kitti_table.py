import sys, json
sys.path.insert(0,'/root/repo')
import bench
print(json.dumps(bench.kitti_table(0), indent=1))
