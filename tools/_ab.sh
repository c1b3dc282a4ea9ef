python -m pytest tests/test_model_gpu.py tests/test_beam_gpu.py tests/test_rules_gpu.py tests/test_fullsize_gpu.py tests/test_edge_gpu.py -x -q 2>&1 | tail -3
python tools/decode_time.py 2>&1 | tail -1
python tools/decode_time.py 2>&1 | tail -1
SATRN_DEC_PROF=1 NB=64 python tools/decode_time.py 2>&1 | grep "dec prof" | tail -11
