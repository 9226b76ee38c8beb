for blk in 256 512; do for rg in 0 1 2; do for mb in 256 512; do
RPE_BLOCK=$blk RPE_REDUCE_GROUPS=$rg RPE_MAX_BLOCKS=$mb timeout 120 python3 scripts/sweep_geometry.py 307200 2>/dev/null | sed "s/^/blk=$blk rg=$rg mb=$mb  /"
done; done; done
