#!/usr/bin/env python3
"""Run every GPU parity check without stopping at failures; write a report to gpurun_out/diag.log.
Usage (GPU box): python tests/gpu_diag.py [substring-filter]"""
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpu_checks as gc  # noqa: E402


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ''
    os.makedirs(os.path.join(gc.ROOT, 'gpurun_out'), exist_ok=True)
    log = open(os.path.join(gc.ROOT, 'gpurun_out', 'diag.log'), 'a')
    npass = nfail = 0
    for fn, a in gc.all_checks():
        label = f'{fn.__name__}{a}'
        if flt and flt not in label:
            continue
        t0 = time.time()
        try:
            r = fn(*a)
            line = f"{'PASS' if r['ok'] else 'FAIL'} {r['name']}: err {r['err']:.3e} (tol {r['tol']:.1e}) {r['note']} [{time.time() - t0:.1f}s]"
            npass += r['ok']
            nfail += not r['ok']
        except BaseException as e:  # noqa: BLE001  (pytest.skip raises an OutcomeException, a BaseException)
            if type(e).__name__ == 'Skipped':
                print(f'SKIP {label}: {e}', flush=True)
                continue
            if isinstance(e, KeyboardInterrupt):
                raise
            nfail += 1
            line = f'ERROR {label}: {type(e).__name__}: {e}\n' + traceback.format_exc(limit=6)
        print(line, flush=True)
        log.write(line + '\n')
        log.flush()
    summary = f'SUMMARY pass={npass} fail={nfail}'
    print(summary)
    log.write(summary + '\n')


if __name__ == '__main__':
    main()
