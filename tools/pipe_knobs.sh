# sweep of the pipelined decoder's shard counts (run on the GPU box): one line per setting
run() { echo "== $*"; env "$@" python tools/decode_time.py 2>&1 | grep pipelined; }
run X=1
run SATRN_KNOBS=pipe_mv_shards=3,pipe_att_shards=10,pipe_xatt_shards=8 
run SATRN_KNOBS=pipe_mv_shards=3,pipe_att_shards=9,pipe_xatt_shards=8,pipe_ln_shards=3 
run SATRN_KNOBS=pipe_att_shards=9,pipe_xatt_shards=5 
run SATRN_KNOBS=pipe_att_shards=8,pipe_xatt_shards=7,pipe_gen_shards=3,pipe_ln_shards=1 
run SATRN_KNOBS=pipe_hist_shards=1,pipe_att_shards=9,pipe_xatt_shards=7 
