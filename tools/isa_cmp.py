import subprocess, sys, hashlib, re, os, tempfile, shutil
O='/opt/rocm/lib/llvm/bin'
def funcs(lib):
    d=tempfile.mkdtemp(); shutil.copy(lib, d+'/lib.so')
    subprocess.run([O+'/llvm-objdump','--offloading','lib.so'],cwd=d,capture_output=True)
    out={}
    for f in os.listdir(d):
        if 'gfx950' not in f: continue
        txt=subprocess.run([O+'/llvm-objdump','-d',d+'/'+f],capture_output=True,text=True).stdout
        cur=None; body=[]
        for line in txt.splitlines():
            m=re.match(r'^[0-9a-f]+ <(.+)>:$',line)
            if m:
                if cur: out[cur]=hashlib.md5('\n'.join(body).encode()).hexdigest()
                cur=m.group(1); body=[]
            elif cur:
                body.append(re.sub(r'//.*$','',re.sub(r'^\s*[0-9a-f]+:\s*','',line)).strip())
        if cur: out[cur]=hashlib.md5('\n'.join(body).encode()).hexdigest()
    return out
a=funcs(sys.argv[1]); b=funcs(sys.argv[2])
hb=set(b.values())
missing=[k for k,v in a.items() if v not in hb]
print(len(a),'old funcs',len(b),'new funcs; old bodies not found in new:',len(missing))
for k in missing[:10]: print('  ',subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip()[:150])
