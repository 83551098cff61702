#!/usr/bin/env python3
"""Traversal diagnostics of one workload: counted iterations, then the library's PT_DEBUG_TAIL report."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PT_DEBUG_TAIL"] = "1"
import __graft_entry__ as entry
pt = entry._load_package()
scenes = importlib.import_module("cuda_path_tracer_amd.scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
sc = scenes.make_scene(name, res=(1920, 1080), depth=8)
tr = pt.Tracer(sc, device=0)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tr.set_option(pt.PT_OPT_BATCH, B)
tr.set_option(pt.PT_OPT_COUNTERS, 1)
tr.trace_range(1, 1 + B, 1)
s = tr.stats()
print("pathBounces", s.pathBounces, "nodes", s.nodesVisited, "tris", s.trisTested, "shadow nodes", s.shadeNodesVisited, "shadow tris", s.shadeTrisTested)
