#!/bin/bash
# Records what the GPU box offers for pinning the oracle to the real reference: a Go toolchain (to build elPrep), an elprep
# binary, samtools/htslib, GATK/Picard, Java.  Output is committed under profiles/ (see DESIGN.md section 6).
echo "== date: $(date -u)"
echo "== host: $(uname -a)"
echo "== cores: $(nproc)   mem: $(free -g | awk '/Mem:/{print $2" GiB"}')"
for t in go gofmt elprep samtools bcftools bgzip htsfile gatk picard java javac node rustc cargo; do
  p=$(command -v $t 2>/dev/null)
  echo "which $t: ${p:-<absent>}"
done
ls -d /usr/local/go /usr/lib/go* /root/go /opt/go* 2>/dev/null || echo "no Go installation directories"
ls /root/reference 2>/dev/null | head -3 || true
[ -d /root/reference ] || echo "/root/reference: <absent>"
find / -xdev \( -name 'elprep*' -o -name 'pargo*' \) -not -path '*/proc/*' -not -path "$GRAFT_REPO_ROOT/*" -not -path '/root/repo/*' 2>/dev/null | head -5
echo "== GOPATH=${GOPATH:-<unset>} GOROOT=${GOROOT:-<unset>}"
rocm-smi --showproductname 2>/dev/null | head -8
