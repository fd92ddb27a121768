"""Static scan of the gfx950 assembly for waits on loads that were only just issued -- the pattern a control-flow merge of a
prefetched value, or a load placed where its result is used, leaves behind:  s_waitcnt vmcnt(n) at most `near` instructions after
the last of a burst of global loads, with n smaller than the number of loads (and stores) issued since the previous wait that
drained them.  Usage: python tools/isa_waits.py <dir with *.s from hipcc -S --cuda-device-only> [kernel-name substring]"""
import glob, re, sys

def kernels(text):
    for m in re.finditer(r'^(_Z\w+):', text, re.M):
        name = m.group(1)
        end = text.find('.Lfunc_end', m.end())
        if end < 0: continue
        body = text[m.end():end]
        ins = [l.strip() for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith('.') and not l.strip().startswith(';')]
        if any(t.startswith('s_endpgm') for t in ins):
            yield name, ins

def scan(ins, near=10):
    hits = []
    pending = 0          # VMEM ops issued since the last vmcnt(0)-like drain (upper bound of what is in flight)
    last_load = None
    burst = 0
    for i, t in enumerate(ins):
        op = t.split()[0]
        if op.startswith(('global_load', 'buffer_load', 'flat_load', 'scratch_load')):
            pending += 1; burst = burst + 1 if last_load is not None and i - last_load < 6 else 1; last_load = i
        elif op.startswith(('global_store', 'buffer_store', 'flat_store', 'scratch_store', 'global_atomic', 'buffer_atomic')):
            pending += 1
        elif op.startswith('s_waitcnt') and 'vmcnt' in t:
            n = int(re.search(r'vmcnt\((\d+)\)', t).group(1))
            if last_load is not None and i - last_load <= near and n < burst:
                hits.append((i, n, burst, i - last_load))
            pending = min(pending, n)
    return hits

if __name__ == "__main__":
    d = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    for f in sorted(glob.glob(d + "/*.s")):
        for name, ins in kernels(open(f).read()):
            if sub and sub not in name: continue
            h = scan(ins)
            if h:
                print("%-22s %-70s %5d instr  %d waits on just-issued loads: %s" % (f.split('/')[-1], name[:70], len(ins), len(h),
                      " ".join("@%d:vmcnt(%d)/burst %d" % (a, b, c) for a, b, c, _ in h[:6])))
