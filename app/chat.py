"""Terminal chat over a speculative-decoding engine.

Same behaviour as the reference's ``app/chatbot.py:21-51`` -- first turn: system prompt + templated user text through
``prefill``; later turns: templated user text through ``append``; each answer streamed to the terminal by
``speculative_decoding``; ``BYE`` ends the conversation, and so does a context that no longer fits
(``validate_status``) -- written as a small state machine so the turn logic can be tested without a terminal or a GPU.

    python -m app.chat --configuration configs/static_70b_awq_on_device.yaml
"""
from __future__ import annotations

import argparse
import enum
from dataclasses import dataclass, field
from typing import Callable


class TurnResult(enum.Enum):
    ANSWERED = "answered"
    GOODBYE = "goodbye"                 # the user typed the stop word
    OUT_OF_CONTEXT = "out_of_context"   # the prompt did not fit, or nothing fits after this answer


@dataclass
class ChatSession:
    """One conversation on one engine.  `engine` needs prefill / append / speculative_decoding / validate_status / reset."""
    engine: object
    system_prompt: str
    user_template: str
    generation_length: int = 256
    max_turns: int = 16
    stop_word: str = "BYE"
    turns_done: int = 0
    transcript: list = field(default_factory=list)        # (user text, (tokens, seconds, target steps)) per answered turn

    def say(self, user_text: str) -> TurnResult:
        if user_text.strip() == self.stop_word:
            return TurnResult.GOODBYE
        framed = self.user_template.format(user_text)
        if self.turns_done == 0:
            fits = self.engine.prefill(self.system_prompt + framed)
        else:
            fits = self.engine.append(framed)
        if not fits:                                            # overflow: the engine returns False, it never raises
            return TurnResult.OUT_OF_CONTEXT
        stats = self.engine.speculative_decoding(max_new_tokens=self.generation_length)
        self.transcript.append((user_text, stats))
        self.turns_done += 1
        return TurnResult.ANSWERED if self.engine.validate_status() else TurnResult.OUT_OF_CONTEXT

    def run(self, read: Callable[[str], str] = input, tell: Callable[[str], None] = print) -> TurnResult:
        """The interactive loop; returns why it ended."""
        from umbrella_amd.utils import TextColors
        tell(TextColors.colorize("Chat started -- type " + self.stop_word + " to leave", "cyan"))
        last = TurnResult.ANSWERED
        while self.turns_done < self.max_turns:
            text = read(TextColors.colorize("User: ", "blue"))
            print(TextColors.colorize("Assistant:", "blue"), end=" ", flush=True)
            last = self.say(text)
            if last is TurnResult.GOODBYE:
                tell(TextColors.colorize("Bye.", "cyan"))
                break
            if last is TurnResult.OUT_OF_CONTEXT:
                tell(TextColors.colorize("The conversation no longer fits the reserved context. Bye.", "cyan"))
                break
        return last


def session_from_config(config: dict, device: str = "cuda:0") -> ChatSession:
    """Reference-style configuration (``configs/*.json`` / ``*.yaml``) -> an initialised engine inside a ChatSession."""
    from umbrella_amd.speculation.auto_engine import AutoEngine
    from umbrella_amd.templates import Prompts, SysPrompts
    config = dict(config)
    gen_len = config.pop("generation_length", 256)
    max_turns = config.pop("max_turns", 16)
    template = config.pop("template", "meta-llama3")
    engine = AutoEngine.from_config(device, **config)
    engine.initialize()
    return ChatSession(engine, SysPrompts[template], Prompts[template], generation_length=gen_len, max_turns=max_turns)


def main(argv=None):
    from umbrella_amd.utils import load_config
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--configuration", default="configs/static_70b_awq_on_device.yaml")
    ap.add_argument("--device", default="cuda:0")
    args = ap.parse_args(argv)
    session_from_config(load_config(args.configuration), args.device).run()


if __name__ == "__main__":
    main()
