import csv, sys
allrows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in allrows if 'naive_conv' not in r['Name']]
naive = [r for r in allrows if 'naive_conv' in r['Name']]
steps = int(sys.argv[2])
def c(n):
    if 'pointwise_' in n or 'conv3x3' in n: return 'OUR 1x1 convolutions (fwd+stats / dgrad / wgrad) + 3x3 wgrad'
    if 'adamw_multi' in n: return 'OUR AdamW'
    if 'upcat' in n or 'graph_fill' in n or 'graph_copy' in n: return 'OUR HIP kernels'
    if 'igemm_fwd' in n or 'conv_fwd' in n or 'ConvFwd' in n or 'gemm_fwd' in n: return 'conv fwd'
    if 'igemm_bwd' in n or 'conv_bwd_data' in n: return 'conv bwd-data'
    if 'igemm_wrw' in n or 'conv_bwd_weight' in n or 'wrw' in n.lower(): return 'conv bwd-weight'
    if 'SubTensorOp' in n: return 'MIOpen tensor-op (wrw zero/cast)'
    if 'bn_stats' in n or 'bn_apply' in n or 'bn_bwd' in n or 'bn_infer' in n: return 'OUR fused BN+act'
    if 'maxpool5' in n: return 'OUR maxpool5'
    if 'BatchNorm' in n or 'batch_norm' in n: return 'batchnorm'
    if 'silu' in n: return 'silu'
    if 'max_pool' in n: return 'maxpool'
    if 'reduce_kernel' in n: return 'reduce'
    if 'copy' in n.lower(): return 'copies/casts'
    if 'yolo' in n or 'hd_' in n or 'nms' in n or 'iou' in n: return 'OUR HIP kernels'
    if 'multi_tensor' in n or 'FusedOpti' in n: return 'optimizer/foreach'
    if 'CatArray' in n: return 'cat'
    if 'upsample' in n: return 'upsample'
    if 'fillBuffer' in n: return 'memset'
    if 'Cijk' in n or 'gemm' in n.lower(): return 'gemm-like (conv via gemm)'
    return 'other'
cat = {}
for r in rows:
    k = c(r['Name']); cat[k] = cat.get(k, 0) + float(r['TotalDurationNs'])
tot = sum(cat.values())
print(f"total kernel time {tot/1e6/steps:.3f} ms/step over {steps} steps")
for k, v in sorted(cat.items(), key=lambda x: -x[1]): print(f"  {k:40s} {v/1e6/steps:7.3f} ms/step")
print("top kernels:")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print(f"  {r['Name'][:100]:100s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f} tot_ms/step={float(r['TotalDurationNs'])/1e6/steps:.3f}")
if naive:
    print(f"excluded: {sum(int(r['Calls']) for r in naive)} calls of MIOpen's find-mode reference kernels (naive_conv_*, the solver search of "
          f"the first warm-up step of the process), {sum(float(r['TotalDurationNs']) for r in naive)/1e6:.1f} ms in total")
