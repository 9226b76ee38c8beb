for blk in 256 512; do for mb in 256 512 1024; do
echo "blk=$blk mb=$mb"; RPE_BLOCK=$blk RPE_MAX_BLOCKS=$mb timeout 120 python3 scripts/nl_round_probe.py 2>/dev/null | grep 10000000 | cut -c1-120
done; done
