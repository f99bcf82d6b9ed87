"""CPU emulation of the index chain of the 32x32x16 four-wave GEMM (tools/probes/gemm_w4.hip: linear_kernel_256w4m): the LDS image the
register staging writes, the fragment addresses, the MFMA operand / result layouts, the two-swap conversion into the eight-wave
accumulator layout, and the bank slots of every ds_read_b128 / ds_write_b128 cycle.  Run before the kernel ever saw a GPU."""
import numpy as np
ROW2=64; OPER2=256*64; STAGE2=2*OPER2
def swz8(row):
    x=(row>>2)&7
    return (x&3) ^ ((x>>2)*3)
# LDS as dict: byte offset (16B granularity) -> (operand, row, kstage, kchunk8)  meaning 8 bf16 k = 8*kchunk8.. of that stage
lds={}
for tid in range(256):
    c8=tid&7; row0=tid>>3
    for j in range(16):
        r=j&7; op = 'x' if j<8 else 'w'
        row=r*32+row0
        base=(c8>>2)*STAGE2 + row0*ROW2 + (((c8&3)^swz8(row0))<<4) + (OPER2 if op=='w' else 0)
        dst=base + r*32*ROW2          # unit_slot 0
        assert dst%16==0 and dst not in lds, (tid,j)
        # content: global piece c8 of row `row`: unit K elements 8*c8..8*c8+7 -> stage (c8>>2), chunk (c8&3)
        lds[dst]=(op,row,c8>>2,c8&3)
assert len(lds)==2*2*256*4
# expected natural layout check: image st, operand, row, position p holds chunk p ^ swz8(row)
for (dst,(op,row,st,ch)) in lds.items():
    off=dst - st*STAGE2 - (OPER2 if op=='w' else 0)
    assert off//ROW2==row and (off%ROW2)//16 == (ch ^ swz8(row)), (dst,op,row,st,ch)
ok=True
for wave in range(4):
  wm=wave>>1; wn2=wave&1
  for stage in range(2):
    buf=stage*STAGE2
    # D accumulators in "new" layout: newD[tm][tn][lane][reg] = (token,row feature) pair set; verify operands first
    for lane in range(64):
        c32=lane&31; h2=lane>>5
        off_x0=(wm*128+c32)*ROW2
        wrow=wn2*128 + 16*((c32>>2)&1) + 4*(c32>>3) + (c32&3)
        off_w0=OPER2+wrow*ROW2
        sx=swz8(wm*128+c32); sw=swz8(wrow)
        for kh in range(2):
            px=((2*kh+h2)^sx)<<4; pw=((2*kh+h2)^sw)<<4
            for t in range(4):
                e=lds[buf+off_x0+t*32*ROW2+px]
                assert e==('x', wm*128+32*t+c32, stage, 2*kh+h2), (e,lane,kh,t)
                e=lds[buf+off_w0+t*32*ROW2+pw]
                assert e==('w', wrow+32*t, stage, 2*kh+h2), (e,lane,kh,t)
  # MFMA semantic: A lane l: row i=l&31, k-half l>>5 (8 k each); B same for col j. D reg 4a+b of lane l: row 8a+4(l>>5)+b, col l&31.
  # A row i of block tn = weight row wn2*128 + 32*tn + 16*((i>>2)&1) + 4*(i>>3) + (i&3); B col j of tm = token wm*128+32tm+j
  def feat(tn,i): return wn2*128 + 32*tn + 16*((i>>2)&1) + 4*(i>>3) + (i&3)
  new={}   # (tm,tn,lane,reg) -> (token,feature)
  for tm in range(4):
    for tn in range(4):
      for lane in range(64):
        for reg in range(16):
            a=reg>>2; b=reg&3
            i=8*a+4*(lane>>5)+b; j=lane&31
            new[(tm,tn,lane,reg)]=(wm*128+32*tm+j, feat(tn,i))
  def swap32(X,Y):   # X'=[Xlo,Ylo], Y'=[Xhi,Yhi]
      return X[:32]+Y[:32], X[32:]+Y[32:]
  def swap16(A,B):   # A'=[a0 b0 a2 b2], B'=[a1 b1 a3 b3]
      r=lambda V,k: V[16*k:16*k+16]
      return r(A,0)+r(B,0)+r(A,2)+r(B,2), r(A,1)+r(B,1)+r(A,3)+r(B,3)
  for hh in range(2):
    for tm in range(4):
      for a in range(4):
        for b in range(4):
            X=[new[(tm,2*hh,l,4*a+b)] for l in range(64)]
            Y=[new[(tm,2*hh+1,l,4*a+b)] for l in range(64)]
            X1,Y1=swap32(X,Y); X2,Y2=swap16(X1,Y1)
            for e,V in ((0,X2),(1,Y2)):
                fm=2*tm+e; fn=a
                for lane in range(64):
                    i16=lane&15; g=lane>>4
                    wn=2*wn2+hh
                    want=(wm*128+16*fm+i16, wn*64+16*g+4*fn+b)
                    if V[lane]!=want:
                        ok=False; print("MISMATCH",wave,hh,tm,a,b,e,lane,V[lane],want); raise SystemExit
print("emulation OK" if ok else "FAILED")
# bank-conflict check for ds_read_b128 groups
groups=[[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
groups+= [[l+32 for l in g] for g in groups]
for op in ('x','w'):
  worst=0
  for g in groups:
    slots=set()
    for lane in g:
        c32=lane&31; h2=lane>>5
        row = c32 if op=='x' else 16*((c32>>2)&1)+4*(c32>>3)+(c32&3)
        addr=row*ROW2 + ((h2 ^ swz8(row))<<4)
        slots.add((addr%256)//16)
    worst=max(worst,16-len(slots))
  print(op,"read conflicts (16 - distinct slots):",worst)
# write conflicts
worst=0
for wave in range(4):
  for g in groups:
    slots=set()
    for lane in g:
        tid=wave*64+lane; c8=tid&7; row0=tid>>3
        addr=(c8>>2)*STAGE2 + row0*ROW2 + (((c8&3)^swz8(row0))<<4)
        slots.add((addr%256)//16)
    worst=max(worst,16-len(slots))
print("write conflicts:",worst)
