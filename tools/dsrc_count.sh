for cfg in "SFM_DSRC_MARGIN=12" "SFM_DSRC_MARGIN=24" "SFM_DSRC_MARGIN=12 SFM_DSRC_LDS_KB=78"; do
 for wl in cfg3_edge cfg3_smooth_disp; do echo "== $cfg $wl"; env $cfg SFM_DSRC_COUNT=1 timeout -k 10 100 python tools/dsrc_once.py $wl 6 2>&1 | grep "dsrc counters"; done
done
