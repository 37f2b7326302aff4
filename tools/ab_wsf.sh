# the fp32 1x1 layers of YOLOv4-L on the weight-stationary kernel / the generic tile / the wide tile (product library)
python tools/conv_bench.py --filter "k1s1" --tiles 9,6,11 --reps 7 2>&1 | grep -v amdgpu
