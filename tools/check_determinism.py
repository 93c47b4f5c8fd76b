import sys, numpy as np
sys.path.insert(0, '.')
import bench, ndpp_amd
wl = bench.make_workload(32768, 6)
p = ndpp_amd.Params.default(6, wl["M"])
args = (wl["A"], wl["kT"], 1e300, 0.0, wl["ein"], wl["row_lo"], wl["w_hi"], wl["f_tab"], wl["bins"])
a, sa = ndpp_amd.elastic_leg_batch(p, *args)
b, sb = ndpp_amd.elastic_leg_batch(p, *args)
# and as two half-size shards (different chunk composition, split mode at other levels)
h = 16384
c1, _ = ndpp_amd.elastic_leg_batch(p, wl["A"], wl["kT"], 1e300, 0.0, wl["ein"][:h], wl["row_lo"][:h], wl["w_hi"][:h], wl["f_tab"], wl["bins"])
c2, _ = ndpp_amd.elastic_leg_batch(p, wl["A"], wl["kT"], 1e300, 0.0, wl["ein"][h:], wl["row_lo"][h:], wl["w_hi"][h:], wl["f_tab"], wl["bins"])
# a tiny shard goes through the split walk on every level
t, _ = ndpp_amd.elastic_leg_batch(p, wl["A"], wl["kT"], 1e300, 0.0, wl["ein"][1000:1064], wl["row_lo"][1000:1064], wl["w_hi"][1000:1064], wl["f_tab"], wl["bins"])
print("rerun identical:", np.array_equal(a, b), " halves identical:", np.array_equal(a, np.concatenate([c1, c2])),
      " 64-point shard identical:", np.array_equal(a[1000:1064], t), " status:", int(sa.sum()))
