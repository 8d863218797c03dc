import os, sys, time, torch
sys.path.insert(0, '.')
import bench
from timeautodiff_amd import DIFF, ops
dev = torch.device('cuda:0')
cx = bench.Ctx(dev, 1, 0) if hasattr(bench, 'Ctx') else None
for name in (sys.argv[1:] or ['energy_T24_H200']):
    wl = bench.WORKLOADS[name]
    schema = wl['schema']
    F_ = schema['n_bins'] + schema['n_cats'] + schema['n_nums']
    torch.manual_seed(1)
    latent, time_info, cond = bench.synth(4096, 1, dev, schema, wl['T'])
    model = DIFF.BiRNN_score(bench.K, F_, wl['H'], 2, 100, 8, bench.E, schema['n_bins'], schema['n_cats'], schema['n_nums'], schema['cards'])
    model.set_compute_dtype('bf16')
    Bs = 1024
    tg = torch.linspace(0, 1, wl['T'], device=dev).view(1, -1, 1).repeat(Bs, 1, 1)
    run = lambda: DIFF.sample(tg, Bs, wl['T'], bench.K, 100, model, time_info[:Bs], cond[:Bs], None, noise_source='device', streams=int(os.environ.get('STREAMS', '2')))
    for _ in range(2): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): run()
    torch.cuda.synchronize(); print(name, F_, (time.perf_counter() - t0) / 3 * 1e3, 'ms')
    with ops.LaunchTrace() as tr:
        run()
    agg = {}
    for key, kern, e0, e1, cost in tr.records:
        torch.cuda.synchronize()
        a = agg.setdefault((key[0], kern), [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1)
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]: print('   ', k, v[0], round(v[1], 2), 'ms')
