"""Development harness: tile-engine self test + golden parity + a first timing."""
import glob, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from bayesian_optimization_amd import GP
from oracle import gp_oracle as o

gp = GP(0)
print('selftest n=256 err', gp.selftest_gemm(256), flush=True)
err, rate = gp.selftest_gemm(4096, rate=True)
print('selftest n=4096 err %g rate %.2f TFLOP/s' % (err, rate), flush=True)

e = lambda a, b: np.max(np.abs(a - b)) / np.max(np.abs(b))
for f in sorted(glob.glob(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'g[1-5]*.npz'))):
    g = np.load(f)
    try:
        gp.fit(g['xTrain'], g['yTrain'], g['theta'], float(g['sigmaY']))
    except np.linalg.LinAlgError as ex:
        print(os.path.basename(f), 'LinAlgError', ex, 'ref raised', int(g['raised_linalgerror']))
        continue
    mu, var, ei, idx, eim, mum = gp.predict_ei(g['xTest'], float(g['kappa']))
    print('%-28s nll %.1e mu %.1e var %.1e ei %.1e argmax %d/%d' % (os.path.basename(f), abs(gp.nll - g['nll']) / abs(g['nll']),
          e(mu, g['Mu']), e(var, g['var']), e(ei, g['ei']), idx, int(g['argmax'])), flush=True)

# first timing at the cfg2 shape
n, d, m = 4096, 4, 1 << 20
X = o.splitmix64_uniform(1, n * d).reshape(n, d)
y = np.sin(3 * X.sum(1)) + 0.1 * (o.splitmix64_uniform(11, n) - 0.5)
Xs = o.splitmix64_uniform(2, m * d).reshape(m, d)
t0 = time.time(); gp.fit(X, y, np.full(d, 0.35), 0.1); t1 = time.time()
print('fit n=4096: wall %.3fs' % (t1 - t0), {k: round(v, 3) for k, v in gp.timings().items() if 'ms' in k and v}, flush=True)
gp.set_candidates(Xs)
for it in range(2):
    t0 = time.time(); mm = gp.posterior(); idx, best = gp.acquire(); t1 = time.time()
    tm = gp.timings()
    print('posterior+acquire m=1M: wall %.3fs cand/s %.3e' % (t1 - t0, m / (t1 - t0)), {k: round(v, 3) for k, v in tm.items() if v}, flush=True)
    print('trmm TFLOP/s', tm['trmm_flops'] / (tm['trmm_ms'] * 1e-3) / 1e12)
# parity on a subsample vs oracle
st = o.fit(X, y, np.full(d, 0.35), 0.1)
sub = np.arange(0, m, 257)[:4096]
mu_o, var_o = o.predict(st, Xs[sub])
mu, var = gp.read()
print('cfg2 subsample parity mu %.2e var %.2e theta1 %.3e nll %.3e' % (e(mu[sub], mu_o), e(var[sub], var_o),
      abs(gp.theta1 - st['theta1']) / st['theta1'], abs(gp.nll - st['nll']) / abs(st['nll'])))
