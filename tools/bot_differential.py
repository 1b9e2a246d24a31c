#!/usr/bin/env python3
"""One-off differential sweep (build container only): the reference's own `Bot` (babyai/bot.py, unmodified on the shim)
next to the host build of babyai_amd/csrc/bbai_bot.hpp on every registered level, fresh seeds, pure and perturbed
(the bot is told the action really taken).  Prints one line per level and a MISMATCH line for any decision that
differs; DESIGN.md section 9 quotes the totals.

    PYTHONDONTWRITEBYTECODE=1 timeout 3000 python tools/bot_differential.py [seed_base] [seeds_per_level] [p_random]
"""
import os, sys, signal
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
SEED_BASE = int(sys.argv[1]) if len(sys.argv) > 1 else 123000
N_SEEDS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
P_RANDOM = float(sys.argv[3]) if len(sys.argv) > 3 else 0.15
from oracle import refenv
refenv.import_reference()
from babyai.bot import Bot
from babyai.levels import level_dict
from babyai_amd.levels import make_cfg
from hostsim_util import HostBot, HostEnv
class TO(BaseException): pass
def on_alarm(s,f): raise TO()
rng = np.random.RandomState(99)
total=0; bad=0; dead=0
for level in sorted(level_dict):
    for k in range(N_SEEDS):
        seed = SEED_BASE + 101*k
        ref = level_dict[level]()
        if hasattr(ref,'locked_room'): ref.locked_room=None
        ref.seed(seed); ref.reset()
        env = HostEnv(make_cfg(level), seed); env.reset()
        rbot, hbot = Bot(ref), HostBot(env); first, last = True, None
        p = 0.0 if k%2==0 else P_RANDOM
        for t in range(160):
            try:
                signal.signal(signal.SIGALRM,on_alarm); signal.setitimer(signal.ITIMER_REAL,3.0)
                try: want=int(rbot.replan(last))
                finally: signal.setitimer(signal.ITIMER_REAL,0)
            except BaseException as e:
                if isinstance(e,KeyboardInterrupt): raise
                want=None
            got=hbot.decide(first,last); first=False; total+=1
            if got!=want:
                bad+=1; print("MISMATCH",level,seed,t,got,want,flush=True); break
            if want is None:
                dead+=1
                a=int(rng.randint(0,7))
                # reference driver: bot gone until the episode ends -> stop comparing this episode
                while True:
                    _,r,d,_=ref.step(a); _,hr,hd=env.step(a)
                    assert bool(d)==bool(hd)
                    if d: break
                    a=int(rng.randint(0,7))
                ref.reset(); env.reset(); rbot,hbot=Bot(ref),HostBot(env); first,last=True,None
                continue
            a = want if rng.rand()>=p else int(rng.randint(0,7))
            last=a
            _,r,d,_=ref.step(a); _,hr,hd=env.step(a)
            assert bool(d)==bool(hd) and np.float32(r)==hr, (level,seed,t)
            if d:
                ref.reset(); env.reset(); rbot,hbot=Bot(ref),HostBot(env); first,last=True,None
    print(level,"ok total",total,"dead",dead,"bad",bad,flush=True)
print("DONE total",total,"dead",dead,"bad",bad)
