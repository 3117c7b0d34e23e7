#!/bin/bash
# round 5, GPU call 51: the new knob test, with the other re-trace tests
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "critical or retrace or wave_mates or rerun or knobs" 2>&1 | tail -n 6
