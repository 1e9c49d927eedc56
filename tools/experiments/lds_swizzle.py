import itertools
GROUPS=[list(range(0,4))+list(range(12,16))+list(range(20,28)), list(range(4,12))+list(range(16,20))+list(range(28,32))]
GROUPS+= [[l+32 for l in g] for g in GROUPS]
def conflicts(addr):  # addr: fn lane->byte address (16B aligned); returns max ways
    worst=1
    for g in GROUPS:
        banks={}
        for l in g:
            a=addr(l)
            for b in range(4):
                banks.setdefault(((a//4)+b)%64,set()).add(a)
        worst=max(worst,max(len(s) for s in banks.values()))
    return worst
# Cin=64: 128B rows, 8 chunks; ks in 0,1
for name,f in [("xor7",lambda c,px:c^(px&7)),("none",lambda c,px:c)]:
    print("cin64",name,[conflicts(lambda l,ks=ks:(l&15)*128+f(4*ks+(l>>4),l&15)*16) for ks in (0,1)])
# Cin=32: 64B rows, 4 chunks
best=[]
for perm in itertools.product(range(4),repeat=4):
    f=lambda c,px:c^perm[(px>>2)&3]
    w=conflicts(lambda l:(l&15)*64+f(l>>4,l&15)*16)
    if w==1: best.append(perm)
print("cin32 perms conflict-free:",best[:10])
for name,f in [("none",lambda c,px:c),("px>>1",lambda c,px:c^((px>>1)&3)),("px>>2",lambda c,px:c^((px>>2)&3))]:
    print("cin32",name,conflicts(lambda l:(l&15)*64+f(l>>4,l&15)*16))
# Cin=16: 32B rows, 2 chunks, l4&1
print("cin16",conflicts(lambda l:(l&15)*32+((l>>4)&1)*16))
# at reads: px pitch 32B: lane reads (P+l15+tapoff(slot))*32+16*(l4&1), slot=l4>>1
for d in range(0,40):
    w=conflicts(lambda l:((l&15)+ (d if (l>>5) else 0))*32+16*((l>>4)&1))
    if w>1: print("at read slotdelta",d,w)
