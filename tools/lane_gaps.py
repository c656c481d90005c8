"""Where does the main lane of a train step WAIT?  (GPU box; no profiler attached.)

Walks the step's launch plans on their lanes with a timing event in front of and behind every launch (capmi_event_create_timed,
recorded on the launch's lane) and one reference event at the start, so every launch gets an absolute start (the moment its lane
was ready for it: everything before it on that lane has ended AND every event it waits for has been recorded) and end.  For
each lane it prints busy time and idle time, and for the largest idle intervals of the main lane the plan rows between the two
launches: a `wait` row whose event was recorded (on the other lane) AFTER the main lane became free is the cause of that gap.

    python tools/lane_gaps.py [--config 1|3] [--top 40]
"""
import argparse, ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from myimagecaptioningmodel_amd import _lib, default_cfg
from myimagecaptioningmodel_amd._lib import PtrSlot
from myimagecaptioningmodel_amd.model import CaptionEngine


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--top', type=int, default=40)
    ap.add_argument('--batch', type=int, default=64)
    args = ap.parse_args()
    B = args.batch
    cfg = default_cfg(batch_size=B, sample_count=0, **bench.WORKLOAD)
    eng = CaptionEngine(cfg, device='cuda:0', use_graph=False)
    image, cap = bench.synthetic_batch(B, cfg, 1234)
    image_d, cap_d = torch.as_tensor(image).cuda(), torch.as_tensor(cap).cuda()
    for _ in range(5):
        eng.train_step(image_d, cap_d)
    torch.cuda.synchronize()
    prog = eng._train[B]
    L = _lib.lib()
    stream = eng._stream()

    def ev():
        e = ctypes.c_void_p()
        assert L.capmi_event_create_timed(ctypes.byref(e)) == 0
        return e
    rows = []           # (kind, lane, name, key, ev_a, ev_b)
    ref = ev()
    torch.cuda.synchronize()
    L.capmi_event_record(ref, stream)
    for plan in (prog['fwd'], prog['bwd_opt']):
        c = plan._compiled.get(('py', len(plan.calls))) or plan._compile(True)
        plan._compiled[('py', len(plan.calls))] = c
        ptrs = {0: stream}
        ptrs.update({l: s.value for l, s in c['side']['streams'].items()})
        others = [l for l in c['used'] if l]
        L.capmi_event_record(c['side']['fork'], stream)
        for l in others:
            L.capmi_stream_wait_event(ptrs[l], c['side']['fork'])
        events, seen = c['events'], set()
        for fn, name, a in plan.calls:
            if fn is None:
                key, lane = a
                if name == 'record':
                    seen.add(key)
                    L.capmi_event_record(events[key], ptrs[lane])
                    t = ev()
                    L.capmi_event_record(t, ptrs[lane])          # when the record was reached on its lane
                    rows.append(('record', lane, name, key, t, t))
                elif key in seen:
                    L.capmi_stream_wait_event(ptrs[lane], events[key])
                    rows.append(('wait', lane, name, key, None, None))
                continue
            lane = getattr(fn, 'lane', 0)
            ea, eb = ev(), ev()
            L.capmi_event_record(ea, ptrs[lane])
            rc = getattr(fn, 'fn', fn)(*[x.value if isinstance(x, PtrSlot) else x for x in a], ptrs[lane])
            assert rc == 0, (name, _lib.last_error())
            L.capmi_event_record(eb, ptrs[lane])
            rows.append(('launch', lane, name, None, ea, eb))
        for l in others:
            L.capmi_event_record(c['side']['join'][l], ptrs[l])
            L.capmi_stream_wait_event(stream, c['side']['join'][l])
    torch.cuda.synchronize()
    ms = ctypes.c_float(0.0)

    def at(e):
        assert L.capmi_event_elapsed_ms(ref, e, ctypes.byref(ms)) == 0
        return ms.value * 1e3
    T = [(k, lane, name, key, at(a) if a is not None else None, at(b) if b is not None else None) for k, lane, name, key, a, b in rows]
    rec_time = {key: ta for k, lane, name, key, ta, tb in T if k == 'record'}
    end = max(tb for k, lane, name, key, ta, tb in T if k == 'launch')
    print('step: %.1f us from the reference event to the last end (the event pairs themselves cost the queues a few us per launch)' % end)
    for lane in sorted({r[1] for r in T if r[0] == 'launch'}):
        ls = [r for r in T if r[0] == 'launch' and r[1] == lane]
        busy = sum(r[5] - r[4] for r in ls)
        print('lane %d: %d launches, busy %.1f us, first start %.1f, last end %.1f' % (lane, len(ls), busy, ls[0][4], ls[-1][5]))
    # gaps of the main lane: between the end of launch i-1 and the start event of launch i lie only waits / records
    gaps = []
    prev_end, between = None, []
    for i, r in enumerate(T):
        k, lane, name, key, ta, tb = r
        if lane != 0:
            continue
        if k == 'launch':
            if prev_end is not None:
                # `ta` is recorded BEHIND the waits in front of the launch: the lane idles from prev_end to ta (+ the launch's own dispatch)
                gaps.append((ta - prev_end, prev_name, name, list(between), prev_end, ta))
            prev_end, prev_name, between = tb, name, []
        elif k == 'wait':
            between.append(('wait', key, rec_time.get(key)))
        else:
            between.append(('record', key, ta))
    tot = sum(g[0] for g in gaps)
    print('main lane: %.1f us idle in %d intervals between launches; intervals > 3 us: %d (%.1f us)' % (
        tot, len(gaps), sum(1 for g in gaps if g[0] > 3), sum(g[0] for g in gaps if g[0] > 3)))
    binding = 0.0
    print('largest intervals (us idle | after -> before | rows in between: a wait is BINDING when its event was recorded after the lane fell free):')
    for g in sorted(gaps, key=lambda x: -x[0])[:args.top]:
        notes = []
        for kind, key, t in g[3]:
            if kind == 'wait':
                late = (t is not None and t > g[4])
                notes.append('wait %s%s' % ('/'.join(str(x) for x in key), ' BINDING (recorded %.1f us after the lane fell free)' % (t - g[4]) if late else ''))
            else:
                notes.append('record %s' % '/'.join(str(x) for x in key))
        print('  %7.1f | %s -> %s | %s' % (g[0], g[1].replace('capmi_', ''), g[2].replace('capmi_', ''), '; '.join(notes)))
    for g in gaps:
        if any(kind == 'wait' and t is not None and t > g[4] for kind, key, t in g[3]):
            binding += g[0]
    print('idle time of intervals with a binding wait: %.1f us; with a record but no binding wait: %.1f us; with nothing in between: %.1f us' % (
        binding, sum(g[0] for g in gaps if g[3] and not any(kind == 'wait' and t is not None and t > g[4] for kind, key, t in g[3])),
        sum(g[0] for g in gaps if not g[3])))


if __name__ == '__main__':
    main()
