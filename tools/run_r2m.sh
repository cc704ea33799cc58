cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2m
for w in 16 64; do
bash tools/kstats.sh r2m vgg512_w$w DN_PP_WANT=$w -- --model ssd512_vgg16 --batch 32 | grep -E "select|merge|tau|softmax|sum"
done
bash tools/kstats.sh r2m vgg300 -- --model ssd300_vgg16 --batch 64 | grep -E "select|merge|tau|softmax|sum"
