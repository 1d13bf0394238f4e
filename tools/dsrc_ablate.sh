# development: sweeps of the d_src scatter launch (SFM_DSRC_* knobs of sfm_loss.hip / sfm_loss_dsrc.hip); arguments = one knob set per run
run() { echo "== $*"; env "$@" timeout -k 10 120 python tools/dsrc_time.py cfg3_edge cfg3_smooth_disp --only 2>&1 | grep d_src; }
for cfg in "$@"; do run $cfg; done
