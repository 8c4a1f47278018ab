bash tools/dev/ab1.sh c3-1d 100000000 cur notally noresample nothing
for k in 1 2 3; do echo "blocks per CU $k"; JB_TRANSPORT_BLOCKS_PER_CU=$k bash tools/dev/ab1.sh c3-1d 100000000 cur | tail -1; done
echo "c3 3-D"; bash tools/dev/ab1.sh c3 100000000 cur notally noresample nothing
for k in 2 3; do echo "blocks per CU $k"; JB_TRANSPORT_BLOCKS_PER_CU=$k bash tools/dev/ab1.sh c3 100000000 cur | tail -1; done
