#!/bin/bash
# Ten copies of the probe at once, different sizes, 40 s each; mode 0 = the library's setters (stream sync, blocking copy on the
# null stream, kernel on a non-blocking stream right behind), mode 1 = the copy on the kernel's own stream.
#   hipcc --offload-arch=gfx950 -O2 -o /tmp/h2d_order_probe profiles/experiments/h2d_order_probe.hip && bash profiles/experiments/h2d_order_probe.sh
for mode in 0 1; do
  for w in 16 64 256 1024 4096 16384 65536 262144 45 69; do /tmp/h2d_order_probe 40 $w $mode & done
  wait
done
