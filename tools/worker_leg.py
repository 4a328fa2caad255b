#!/usr/bin/env python3
"""bench.py's extra_worker leg alone (single-consumer pool-shaped loop, PNG included): worker_leg.py [clients] [requests]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdlcm_amd  # noqa
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
r = int(sys.argv[2]) if len(sys.argv) > 2 else 128
print(json.dumps(bench.worker_leg(n_clients=n, n_requests=r), indent=1))
