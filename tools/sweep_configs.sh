for cfg in 4 5; do
for t in "64,4,32,1,1,4,1,1" "64,8,32,1,1,4,1,1" "64,16,32,1,1,4,1,1" "64,8,48,2,1,4,1,1" "48,8,32,1,1,4,1,1" "64,8,32,1,2,4,1,1" "64,6,32,1,1,6,1,1"; do
echo -n "cfg $cfg tune $t: "; NGP_TUNE=$t timeout -k 10 120 python3 tools/render_config.py $cfg 4 2>&1 | tail -1
done; done
