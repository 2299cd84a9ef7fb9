#!/usr/bin/env bash
# Stage 4 (network decoding) of the jatts recipes on the MI355X path.  Sourced by egs/*/tts*/run.sh with the
# reference's variables set: expdir, checkpoint, test_set, token_column, verbose, n_gpus
# (reference egs/jsut/tts1/run.sh:237-260) and, optionally, decode_sets: the csv names under data/ to decode (default: the test set;
# the hificaptain recipes also decode dev_raw_feat, reference egs/hificaptain_jp_female/tts1/run.sh:228).  One difference, additive: n_gpus > 1 is honoured (the reference
# forces 1) -- every rank decodes its own shard of the csv, one process per GPU.
# shellcheck disable=SC2154
stage4_decode() {
    # shellcheck disable=SC2012
    [ -z "${checkpoint}" ] && checkpoint="$(ls -dt "${expdir}"/*.pkl | head -1 || true)"
    [ -n "${checkpoint}" ] || { log "no checkpoint under ${expdir}"; exit 1; }
    outdir="${expdir}/results/$(basename "${checkpoint}" .pkl)"
    # stats: the reference writes stats.h5 (compute_statistics.py:94-103).  Hosts without h5py read the .npz twin that
    # tools/h5stats_to_npz.py makes on a machine that has it.
    local stats="${expdir}/stats.h5"
    if ! "${python}" -c 'import h5py' 2>/dev/null; then
        if [ -e "${expdir}/stats.npz" ]; then stats="${expdir}/stats.npz"
        else log "h5py is not installed and ${expdir}/stats.npz does not exist: run tools/h5stats_to_npz.py ${expdir}/stats.h5 where h5py is available"; exit 1; fi
    fi
    # shellcheck disable=SC2086
    for name in ${decode_sets:-${test_set}}; do
        [ ! -e "${outdir}/${name}" ] && mkdir -p "${outdir}/${name}"
        log "Decoding start. See the progress via ${outdir}/${name}/decode.log."
        local launcher=("${python}" -m jatts_amd.bin.tts_decode)
        if [ "${n_gpus}" -gt 1 ]; then
            launcher=("${python}" -m torch.distributed.run --nnodes=1 --nproc-per-node "${n_gpus}" --master-addr 127.0.0.1
                      --master-port "${master_port:-29517}" -m jatts_amd.bin.tts_decode)
        fi
        "${launcher[@]}" \
            --csv "data/${name}.csv" \
            --stats "${stats}" \
            --token-list "${expdir}/tokens.txt" \
            --token-column "${token_column}" \
            --checkpoint "${checkpoint}" \
            --outdir "${outdir}/${name}" \
            --precision "${precision}" \
            --batch-size "${decode_batch_size}" \
            --n_gpus "${n_gpus}" \
            --verbose "${verbose}" > "${outdir}/${name}/decode.log" 2>&1 || { tail -20 "${outdir}/${name}/decode.log"; exit 1; }
        log "Successfully finished decoding of ${name} set."
    done
    log "Successfully finished decoding."
}
