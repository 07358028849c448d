"""Barrier-epoch model of the half-tile-ring GEMM schedule (csrc/gemm.hip: gemm_tile8_body, round 5).

The kernel's two wave groups (waves 0-3, and waves 4-7 one s_barrier behind) are written out as programs of the events that matter --
request e_n (LDS-DMA of a half-tile), counted wait (all but the youngest five requests of THIS wave have landed), fragment read of e_n,
barrier -- and three properties are checked for 1 .. 8 k-tiles:
  * both groups execute the same number of barriers (no deadlock, no wave left behind);
  * RAW: every wave's covering wait of e_n lies at least one barrier BEFORE any wave's read of e_n;
  * WAR: a request that overwrites the slot of e_(n-8) is issued at least two barriers after every read of e_(n-8)
         (the read is retired by the `s_waitcnt lgkmcnt(0)` behind the phase's first barrier).
Event order: 4T = B0(T), 4T+1 = A0(T), 4T+2 = B1(T), 4T+3 = A1(T); e_n is requested in phase n - 7 (prologue: e_0 .. e_6) and read in phase n - 1.
Run by tests/test_host_logic.py::test_gemm_half_tile_ring_schedule_model; `python tools/sim_gemm_t8_schedule.py` prints the barrier counts."""
def program(group, nk):
    ops = []
    def issue(n): ops.append(('issue', n))
    def read(n): ops.append(('read', n))
    def wait(n): ops.append(('wait', n))        # all events <= n landed (this wave's pieces)
    def bar(): ops.append(('bar',))
    issued = []
    for n in range(7): issue(n); issued.append(n)
    wait(issued[-1] - 5)                       # vmcnt(10): all but the last 5 events
    bar()
    read(0)                                    # B0(0)
    ops.append(('lgk',))
    if group == 1: bar()
    def ktile(T):
        # phase p: read event, issue event, wait, bar, lgk, mma, bar
        reads = [4 * T + 1, 4 * T + 2, 4 * T + 3, 4 * (T + 1) + 0]
        for p in range(4):
            g = 4 * T + p
            read(reads[p])
            issue(g + 7); issued.append(g + 7)
            wait(issued[-1] - 5)
            bar(); ops.append(('lgk',)); ops.append(('mma', g)); bar()
    for T in range(nk): ktile(T)
    if group == 0: bar()
    return ops

def check(nk):
    progs = [program(0, nk), program(1, nk)]
    nb = [sum(1 for o in p if o[0] == 'bar') for p in progs]
    assert nb[0] == nb[1], nb
    info = []
    for p in progs:
        ep = 0; reads = {}; waits = {}; issues = {}
        maxw = -1
        for o in p:
            if o[0] == 'bar': ep += 1
            elif o[0] == 'read': reads[o[1]] = ep
            elif o[0] == 'issue': issues[o[1]] = ep
            elif o[0] == 'wait':
                for n in range(maxw + 1, o[1] + 1): waits[n] = ep
                maxw = max(maxw, o[1])
        info.append((reads, waits, issues))
    live_reads = [n for n in info[0][0] if n < 4 * nk]          # events of real tiles
    for X in range(2):
        for n, er in info[X][0].items():
            if n >= 4 * nk: continue                             # dead prefetch of B0(nk): value unused
            for Y in range(2):
                ew = info[Y][1].get(n)
                assert ew is not None and ew < er, ('RAW', nk, n, X, Y, ew, er)
    for Y in range(2):
        for m, ei in info[Y][2].items():
            n = m - 8
            if n < 0: continue
            for X in range(2):
                er = info[X][0].get(n)
                if er is None: continue
                assert ei >= er + 2, ('WAR', nk, m, n, Y, X, ei, er)
    return nb[0]
if __name__ == '__main__':
    for nk in range(1, 9):
        print(nk, 'k-tiles:', check(nk), 'barriers per wave, RAW / WAR ok')
