"""Eager / graphed x one / two streams of the drop-in call sequence at a GPU-bound batch: losses of 5 steps per mode.
(Eager forwards of the pair stay on the caller's stream; tools/race_hunt.py is the round-5 tool that names differing tensors.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_graphed as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for graph, ab in ((False, False), (False, True), (False, True), (True, True)):
  l, s, k = T._run(graph, ab, steps=5, n_base=n, sz=96)
  print("graph %-5s two-streams %-5s: %s" % (graph, ab, ["%.6e" % v for v in l]))
