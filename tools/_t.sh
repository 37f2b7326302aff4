for a in 0 1 2 4 5 7; do
echo "== ablate $a"
YV4_H16_ABLATE=$a timeout 300 python tools/conv_bench.py --dtype bf16 --tiles 2 2>&1 < /dev/null | grep -E "128->128 k3s1 @76|256->256 k1s1 @38|256->256 k3s1 @38|weighted" | cut -c1-90
done
