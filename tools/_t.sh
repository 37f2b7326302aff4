for i in 1 2; do
for a in 0 8; do
echo "== ablate $a"
YV4_H16_ABLATE=$a timeout 300 python tools/conv_bench.py --dtype bf16 --tiles 2 2>&1 < /dev/null | tail -1
done
done
