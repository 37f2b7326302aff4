#!/bin/bash
# What do the per-launch HIP events of the instrumented steps cost the reported rate?  Same box: bench.py with every 4th
# timed step instrumented (the default), every 20th (one of 20), and the plan alone (tools/two_stream_probe.py: no events,
# no copy of the detections to the host).
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
for E in 4 20; do
echo -n "event-every $E fp32: "; python bench.py --event-every $E --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py roofline.frac roofline.instrumented_steps
echo -n "event-every $E bf16: "; python bench.py --event-every $E --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py roofline.frac roofline.instrumented_steps
echo -n "event-every $E cfg3: "; python bench.py --event-every $E --model yolov4s --size 416 --batch 256 --dtype f16 --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python tools/last_json.py roofline.frac roofline.instrumented_steps
done; done
python tools/two_stream_probe.py --dtype bf16 2>/dev/null | grep "one plan"
python tools/two_stream_probe.py --dtype f16 --model yolov4s --size 416 --batch 256 2>/dev/null | grep "one plan"
