"""VGPR / AGPR / SGPR / scratch / LDS / spill count of every kernel in a built library or object (host only):
   python tools/kernel_resources.py dot_amd/libdotmi.so [filter]
Reads the gfx950 code objects embedded in the file and their AMDGPU metadata notes (llvm-readelf)."""
import re, struct, subprocess, sys, tempfile, os

def code_objects(path):
    data = open(path, 'rb').read()
    idx = 0
    while True:
        i = data.find(b'\x7fELF', idx)
        if i < 0:
            return
        idx = i + 4
        if data[i + 18:i + 20] != b'\xe0\x00':
            continue
        shoff = struct.unpack_from('<Q', data, i + 0x28)[0]
        shentsize, shnum = struct.unpack_from('<HH', data, i + 0x3A)
        yield data[i:i + shoff + shentsize * shnum]

def kernels(path):
    out = []
    for co in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix='.elf', delete=False) as f:
            f.write(co)
        txt = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-readelf', '--notes', f.name], capture_output=True, text=True).stdout
        os.unlink(f.name)
        cur = {}
        for line in txt.split('\n'):
            line = line.strip().lstrip('- ')
            for k in ('.name:', '.vgpr_count:', '.agpr_count:', '.sgpr_count:', '.private_segment_fixed_size:',
                      '.group_segment_fixed_size:', '.vgpr_spill_count:'):
                if line.startswith(k):
                    cur[k] = line.split(':', 1)[1].strip()
            if line.startswith('.wavefront_size:') and '.name:' in cur:
                out.append(cur)
                cur = {}
    names = '\n'.join(k['.name:'] for k in out)
    dem = subprocess.run(['c++filt'], input=names, capture_output=True, text=True).stdout.strip().split('\n')
    rows = []
    for k, d in zip(out, dem):
        d = re.sub(r'\(.*', '', d).replace('void dotmi::', '').replace('dotmi::', '')
        rows.append((d, int(k['.vgpr_count:']), int(k.get('.agpr_count:', 0)), int(k['.sgpr_count:']),
                     int(k['.private_segment_fixed_size:']), int(k['.group_segment_fixed_size:']), int(k.get('.vgpr_spill_count:', 0))))
    return sorted(rows)

if __name__ == '__main__':
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    print('%-70s %5s %5s %5s %8s %7s %6s' % ('kernel', 'vgpr', 'agpr', 'sgpr', 'scratch', 'lds', 'spill'))
    for r in kernels(sys.argv[1]):
        if flt in r[0]:
            print('%-70s %5d %5d %5d %8d %7d %6d' % r)
