"""Vendors the KITTI-00 pose-graph INPUT DATA the hot path consumes.

Run in the build container only (needs /root/reference/data).  Copies data files, no source:
  cc.txt               -- keyframe image ids            (read by kitti_surf.cpp:232-254)
  loopConstraints.txt  -- 118 loop constraints          (read by kitti_surf.cpp:145-205)
  framePoses_kf.txt    -- the 2 header lines + only the 771 keyframe rows of framePoses.txt
                          (kitti_surf.cpp:255-292 keeps exactly these rows)
  gt_kf.txt            -- the 771 keyframe rows of 00.txt (KITTI ground truth; kitti_surf.cpp:1427-1440
                          keeps exactly these rows for the RMSE evaluation)
"""
import os
import shutil

SRC = "/root/reference/data/map000000"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kitti00")

os.makedirs(DST, exist_ok=True)
shutil.copyfile(os.path.join(SRC, "cc.txt"), os.path.join(DST, "cc.txt"))
shutil.copyfile(os.path.join(SRC, "loopConstraints.txt"), os.path.join(DST, "loopConstraints.txt"))
ids = {int(x) for x in open(os.path.join(SRC, "cc.txt")).read().split()}
with open(os.path.join(SRC, "framePoses.txt")) as f, \
        open(os.path.join(DST, "framePoses_kf.txt"), "w") as g:
    lines = f.read().splitlines()
    g.write(lines[0] + "\n" + lines[1] + "\n")
    n = 0
    for ln in lines[2:]:
        if ln.strip() and int(ln.split(",")[0]) in ids:
            g.write(ln + "\n")
            n += 1
# ground truth of the keyframes only (00.txt: 4541 rows of 3x4 Pc2w; kitti_surf.cpp:1164-1190, :1427-1440)
gt = open(os.path.join(SRC, "00.txt")).read().splitlines()
with open(os.path.join(DST, "gt_kf.txt"), "w") as g:
    g.write("% KITTI odometry 00 ground truth rows (3x4 Pc2w, row-major) of the 771 keyframes: image id "
            "+ 12 values; from data/map000000/00.txt (read at kitti_surf.cpp:1164-1190)\n")
    for i in sorted(ids):
        g.write(str(i) + " " + gt[i].strip() + "\n")
print("keyframes", len(ids), "rows kept", n)

# KeyFrame .bin files (read by drawPTAMPoints.cpp:285-332): three single ones for the reader /
# re-anchoring tests, and the first 45 keyframes of the map (1.07 MB) as a connected piece of the real
# bundle-adjustment problem ba_demo is run on (tests/test_ba.py)
KF3 = ["KeyFrame000000.bin", "KeyFrame000011.bin", "KeyFrame000012.bin"]
os.makedirs(os.path.join(DST, "keyframes"), exist_ok=True)
for name in KF3:
    shutil.copyfile(os.path.join(SRC, name), os.path.join(DST, "keyframes", name))
os.makedirs(os.path.join(DST, "keyframes45"), exist_ok=True)
first45 = sorted(n for n in os.listdir(SRC) if n.startswith("KeyFrame") and n.endswith(".bin"))[:45]
for name in first45:
    shutil.copyfile(os.path.join(SRC, name), os.path.join(DST, "keyframes45", name))
print("keyframe files:", len(KF3), "+", len(first45))

# rots.txt: `Rw2i` of every keyframe as the reference wrote it (kitti_surf.cpp:486-496, 6 significant
# digits): the rows of the 45 committed keyframe files pin the pose block of sim3opt_read_keyframe_bin
# to a reference-held number (a rots.txt row is the Rw2c of the .bin, not of
# framePoses.txt)
kept = {int(n[len("KeyFrame"):-len(".bin")]) for n in first45}
with open(os.path.join(SRC, "rots.txt")) as f, open(os.path.join(DST, "rots_kf45.txt"), "w") as g:
    g.write("% rows of data/map000000/rots.txt (image id, Rw2i row-major; written at kitti_surf.cpp:486-496) "
            "for the 45 keyframes under keyframes45/\n")
    nrot = 0
    for ln in f:
        if ln.strip() and int(ln.split()[0]) in kept:
            g.write(ln.strip() + "\n")
            nrot += 1
print("rots rows kept", nrot)
