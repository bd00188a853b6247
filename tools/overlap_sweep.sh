for so in pathtracer_amd/lib/variants/libptx_w64.so pathtracer_amd/lib/variants/libptx_w128.so pathtracer_amd/lib/libptx_mi355x.so; do for o in 0 1; do
n=$(basename $so .so)
PTX_OVERLAP=$o PTX_LIB=$PWD/$so timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ovv.json 2>gpurun_out/ovv.err && python3 -c "
import json
d=json.load(open('gpurun_out/ovv.json'))
print('$n overlap $o: %.1fM samples/s ms/step %.2f extend %.0f shadow %.0f'%(d['value']/1e6,d['ms_per_step'],d['extend_mrays_per_s'],d['shadow_mrays_per_s']))" || tail -2 gpurun_out/ovv.err
done; done
