#!/usr/bin/env bash
# Option parser for the recipe scripts: `--some-name value` sets the (already declared) shell variable `some_name`.
# Written for this repository (the reference recipes source Kaldi's utils/parse_options.sh for the same job).
while [ $# -gt 0 ]; do
    case "$1" in
        --help|-h)
            echo "usage: $0 [--<variable> <value> ...]   (any variable declared above the parse step, e.g. --stage 4 --checkpoint x.pkl)"
            exit 0 ;;
        --*=*)
            _name="${1%%=*}"; _name="${_name#--}"; _name="${_name//-/_}"; _value="${1#*=}"; shift ;;
        --*)
            _name="${1#--}"; _name="${_name//-/_}"
            [ $# -ge 2 ] || { echo "$0: option $1 needs a value" >&2; exit 1; }
            _value="$2"; shift 2 ;;
        *) break ;;
    esac
    if [ -z "${_name:-}" ]; then continue; fi
    if ! declare -p "${_name}" >/dev/null 2>&1; then
        echo "$0: unknown option --${_name//_/-}" >&2; exit 1
    fi
    printf -v "${_name}" '%s' "${_value}"
    unset _name _value
done
