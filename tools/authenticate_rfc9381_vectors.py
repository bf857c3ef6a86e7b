#!/usr/bin/env python3
"""Provenance tool for tests/golden/rfc9381_edwards25519_sha512_tai.json.

ECVRF-EDWARDS25519-SHA512-TAI (RFC 9381 section 5.5, suite_string 0x03) written straight from the RFC text with Python
big ints and hashlib -- it imports nothing from this repository, in particular not the oracle it helps to pin -- and run
over the (SK, alpha) pairs of the RFC's Appendix B.3 examples.  The vectors in the JSON file were written down from
recollection; they were committed because every recalled field matched what this script computes bit for bit (80-byte
proofs, 64-byte outputs).  A mismatch would have meant "discard the recalled vector", never "adjust it".
"""
import hashlib
p = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493
d = (-121665 * pow(121666, p-2, p)) % p
I = pow(2, (p-1)//4, p)
def sha(b): return hashlib.sha512(b).digest()
def inv(x): return pow(x, p-2, p)
def add(P, Q):
    x1,y1=P; x2,y2=Q
    t = d*x1*x2*y1*y2 % p
    return ((x1*y2+y1*x2)*inv(1+t) % p, (y1*y2+x1*x2)*inv(1-t) % p)
def mul(k, P):
    R=(0,1)
    for b in bin(k)[2:]:
        R=add(R,R)
        if b=='1': R=add(R,P)
    return R
def enc(P):
    x,y=P
    return (y | ((x&1)<<255)).to_bytes(32,'little')
def dec(b):
    v=int.from_bytes(b,'little'); sign=v>>255; y=v&((1<<255)-1)
    if y>=p: return None
    x2=(y*y-1)*inv(d*y*y+1)%p
    if x2==0:
        return None if sign else (0,y)
    x=pow(x2,(p+3)//8,p)
    if (x*x-x2)%p!=0: x=x*I%p
    if (x*x-x2)%p!=0: return None
    if (x&1)!=sign: x=p-x
    return (x,y)
By = 4*inv(5)%p
B = dec(By.to_bytes(32,'little'))
def prove(SK, alpha):
    h=sha(SK); a=bytearray(h[:32]); a[0]&=248; a[31]&=127; a[31]|=64
    x=int.from_bytes(a,'little'); Y=mul(x,B); PK=enc(Y)
    # encode_to_curve TAI, salt = PK
    ctr=0
    while True:
        hs=sha(b'\x03\x01'+PK+alpha+bytes([ctr])+b'\x00')
        H=dec(hs[:32])
        if H is not None:
            H=mul(8,H)
            if H!=(0,1): break
        ctr+=1
    hstr=enc(H)
    Gamma=mul(x,H)
    k=int.from_bytes(sha(h[32:]+hstr),'little')%L
    U=mul(k,B); V=mul(k,H)
    cs=sha(b'\x03\x02'+PK+hstr+enc(Gamma)+enc(U)+enc(V)+b'\x00')[:16]
    c=int.from_bytes(cs,'little')
    s=(k+c*x)%L
    pi=enc(Gamma)+cs+s.to_bytes(32,'little')
    beta=sha(b'\x03\x03'+enc(mul(8,Gamma))+b'\x00')
    return dict(PK=PK.hex(),x=x.to_bytes(32,'little').hex(),ctr=ctr,H=hstr.hex(),k=k.to_bytes(32,'little').hex(),U=enc(U).hex(),V=enc(V).hex(),pi=pi.hex(),beta=beta.hex())
if __name__ == "__main__":
    import json, os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "rfc9381_edwards25519_sha512_tai.json")
    bad = 0
    for r in json.load(open(path))["vectors"]:
        out = prove(bytes.fromhex(r["sk"]), bytes.fromhex(r["alpha"]))
        out = {k.lower(): v for k, v in out.items()}
        for k in ("pk", "x", "ctr", "h", "k", "u", "v", "pi", "beta"):
            if k in r:
                ok = r[k] == out[k]
                bad += not ok
                print(r["sk"][:8], k, "MATCH" if ok else "DIFF\n  recalled %s\n  computed %s" % (r[k], out[k]))
    raise SystemExit(1 if bad else 0)
